"""madrona_amd -- MI355X-native many-world ECS task-graph backend.

The product is native: ``madrona_amd/_build/libmadrona_hip.so`` (C ABI in
``include/mwhip.h``; hand-written gfx950 kernels) plus a header overlay
(``madrona_amd/include/madrona``) that simulators written against Madrona's
public API compile against.  This Python package is plumbing around it:

* :mod:`madrona_amd.build`      -- in-tree build of the runtime and simulators
* :mod:`madrona_amd.simlib`     -- ctypes binding of the simulator C API
* :mod:`madrona_amd.tensor`     -- exported device columns as PyTorch-ROCm tensors
* :mod:`madrona_amd.distributed`-- one process per GPU, worlds sharded, RCCL
  all-gather of observation tensors
"""

from .simlib import Simulator, hip_lib_path, ref_lib_path  # noqa: F401

__all__ = ["Simulator", "hip_lib_path", "ref_lib_path"]
