"""ctypes binding for the simulator C API in sims/common/sim_c_api.h.

The same wrapper drives a simulator built against the MI355X HIP backend
(``lib<sim>_hip.so``) and one built against the reference CPU backend
(``oracle/_ref/lib<sim>_ref.so``); only tests/, ``__graft_entry__.smoke`` and
bench.py's ``cpu_baseline`` leg are allowed to load the latter.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

import numpy as np

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_BUILD_DIR = os.path.join(REPO_ROOT, "madrona_amd",
                             os.environ.get("MADRONA_HIP_BUILD_DIR", "_build"))
REF_BUILD_DIR = os.path.join(REPO_ROOT, "oracle", "_ref")

SIM_DTYPES = {
    0: np.uint8, 1: np.int8, 2: np.int16, 3: np.int32, 4: np.int64,
    5: np.float16, 6: np.float32,
}


class SimCreateArgs(C.Structure):
    _fields_ = [
        ("num_worlds", C.c_uint32),
        ("seed", C.c_uint32),
        ("gpu_id", C.c_int32),
        ("num_workers", C.c_uint32),
        ("world_base", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class SimTensorInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("dtype", C.c_int32),
        ("ndim", C.c_int32),
        ("dims", C.c_int64 * 4),
        ("on_device", C.c_int32),
    ]


class SimColumnInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("elem_bytes", C.c_uint32),
        ("is_float", C.c_int32),
    ]


def hip_lib_path(sim: str) -> str:
    return os.path.join(HIP_BUILD_DIR, f"lib{sim}_hip.so")


def ref_lib_path(sim: str, speed: bool = False) -> str:
    suffix = "_ref_speed.so" if speed else "_ref.so"
    return os.path.join(REF_BUILD_DIR, f"lib{sim}{suffix}")


def _bind(lib: C.CDLL) -> None:
    lib.sim_create.restype = C.c_void_p
    lib.sim_create.argtypes = [C.POINTER(SimCreateArgs)]
    lib.sim_destroy.argtypes = [C.c_void_p]
    lib.sim_backend.restype = C.c_char_p
    lib.sim_backend.argtypes = [C.c_void_p]
    lib.sim_step.argtypes = [C.c_void_p, C.c_uint32]
    lib.sim_num_tensors.restype = C.c_uint32
    lib.sim_num_tensors.argtypes = [C.c_void_p]
    lib.sim_tensor_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(SimTensorInfo)]
    lib.sim_tensor_ptr.restype = C.c_void_p
    lib.sim_tensor_ptr.argtypes = [C.c_void_p, C.c_uint32]
    lib.sim_tensor_read.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    lib.sim_tensor_write.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    lib.sim_num_columns.restype = C.c_uint32
    lib.sim_num_columns.argtypes = [C.c_void_p]
    lib.sim_column_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(SimColumnInfo)]
    lib.sim_column_dump.restype = C.c_int64
    lib.sim_column_dump.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                    C.POINTER(C.c_int32)]
    lib.sim_hip_run_taskgraph.restype = C.c_int
    lib.sim_hip_run_taskgraph.argtypes = [C.c_void_p, C.c_uint32]
    lib.sim_column_dump_raw.restype = C.c_int64
    lib.sim_column_dump_raw.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    lib.sim_hip_exec.restype = C.c_void_p
    lib.sim_hip_exec.argtypes = [C.c_void_p]
    lib.sim_hip_render.restype = C.c_int
    lib.sim_hip_render.argtypes = [C.c_void_p]
    lib.sim_hip_render_graph.restype = C.c_uint64
    lib.sim_hip_render_graph.argtypes = [C.c_void_p]
    lib.sim_hip_step_graph.restype = C.c_uint64
    lib.sim_hip_step_graph.argtypes = [C.c_void_p]


class KernelStat(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("node_kind", C.c_uint32),
        ("archetype_id", C.c_uint32),
        ("avg_us", C.c_double),
        ("algo_bytes", C.c_double),
        ("rows", C.c_double),
        ("io_declared", C.c_uint32),
        ("workgroups", C.c_uint32),
        ("node_index", C.c_uint32),
        ("pad_", C.c_uint32),
    ]


def _torch_runtime_first() -> None:
    """PyTorch-ROCm bundles its own HIP / HSA runtime; a process that loads the
    system runtime first (through our libraries) and torch afterwards ends up
    with two HSA runtimes and torch reports "No HIP GPUs are available".
    Loading torch's copy first makes our libraries bind to it (same SONAME)."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def runtime_lib() -> C.CDLL:
    """libmadrona_hip.so (the C ABI of include/mwhip.h)."""
    _torch_runtime_first()
    lib = C.CDLL(os.path.join(HIP_BUILD_DIR, "libmadrona_hip.so"), mode=C.RTLD_GLOBAL)
    lib.mwhip_profile.restype = C.c_int32
    lib.mwhip_profile.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32,
                                  C.POINTER(KernelStat), C.c_uint32]
    lib.mwhip_last_error.restype = C.c_char_p
    lib.mwhip_stream.restype = C.c_void_p
    lib.mwhip_stream.argtypes = [C.c_void_p]
    lib.mwhip_run_async.restype = C.c_int
    lib.mwhip_run_async.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.mwhip_synchronize.restype = C.c_int
    lib.mwhip_synchronize.argtypes = [C.c_void_p]
    lib.mwhip_build_launch_graph_with_pack.restype = C.c_int
    lib.mwhip_build_launch_graph_with_pack.argtypes = [
        C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p),
        C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.mwhip_stream_wait_replays.restype = C.c_int
    lib.mwhip_stream_wait_replays.argtypes = [C.c_void_p, C.c_void_p]
    lib.mwhip_set_input_ring.restype = C.c_int
    lib.mwhip_set_input_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_uint64, C.c_uint32]
    lib.mwhip_mark_window.restype = C.c_int
    lib.mwhip_mark_window.argtypes = [C.c_void_p, C.c_uint32]
    lib.mwhip_pack_rows.restype = C.c_int
    lib.mwhip_pack_rows.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p]
    return lib


class Simulator:
    """One simulator instance behind the C API (either backend)."""

    def __init__(self, lib_path: str, num_worlds: int, seed: int = 5,
                 gpu_id: int = 0, num_workers: int = 1, world_base: int = 0,
                 flags: int = 0):
        if not os.path.exists(lib_path):
            raise FileNotFoundError(
                f"{lib_path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # RTLD_LOCAL: every simulator library defines the same C API (and
        # madronaMWHipUserEntry); they must not interpose on each other
        if lib_path.endswith("_hip.so"):
            _torch_runtime_first()
        self.lib = C.CDLL(lib_path, mode=C.RTLD_LOCAL)
        _bind(self.lib)
        self.num_worlds = num_worlds
        args = SimCreateArgs(num_worlds, seed, gpu_id, num_workers, world_base, flags)
        self.handle = self.lib.sim_create(C.byref(args))
        if not self.handle:
            raise RuntimeError(f"sim_create failed for {lib_path}")
        self.backend = self.lib.sim_backend(self.handle).decode()
        self._async = None
        self._tensor_info: Dict[str, Tuple[int, np.dtype, Tuple[int, ...], bool]] = {}
        for i in range(self.lib.sim_num_tensors(self.handle)):
            info = SimTensorInfo()
            self.lib.sim_tensor_info(self.handle, i, C.byref(info))
            dims = tuple(int(info.dims[k]) for k in range(info.ndim))
            self._tensor_info[info.name.decode()] = (
                i, np.dtype(SIM_DTYPES[info.dtype]), dims, bool(info.on_device))
        self._columns: List[Tuple[str, int, bool]] = []
        for i in range(self.lib.sim_num_columns(self.handle)):
            ci = SimColumnInfo()
            self.lib.sim_column_info(self.handle, i, C.byref(ci))
            self._columns.append((ci.name.decode(), int(ci.elem_bytes), bool(ci.is_float)))

    # -- lifecycle ---------------------------------------------------------
    def close(self) -> None:
        if self.handle:
            self.lib.sim_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def step(self, n: int = 1) -> None:
        self.lib.sim_step(self.handle, n)

    # -- tensors -----------------------------------------------------------
    @property
    def tensor_names(self) -> List[str]:
        return list(self._tensor_info.keys())

    def tensor_meta(self, name: str):
        return self._tensor_info[name]

    def tensor_ptr(self, name: str) -> int:
        return int(self.lib.sim_tensor_ptr(self.handle, self._tensor_info[name][0]))

    def read_tensor(self, name: str) -> np.ndarray:
        idx, dtype, dims, _ = self._tensor_info[name]
        out = np.empty(dims, dtype=dtype)
        rc = self.lib.sim_tensor_read(self.handle, idx, out.ctypes.data, out.nbytes)
        if rc != 0:
            raise RuntimeError(f"sim_tensor_read({name}) -> {rc}")
        return out

    def write_tensor(self, name: str, value: np.ndarray) -> None:
        idx, dtype, dims, _ = self._tensor_info[name]
        arr = np.ascontiguousarray(value, dtype=dtype).reshape(dims)
        rc = self.lib.sim_tensor_write(self.handle, idx, arr.ctypes.data, arr.nbytes)
        if rc != 0:
            raise RuntimeError(f"sim_tensor_write({name}) -> {rc}")

    # -- parity dumps ------------------------------------------------------
    @property
    def columns(self) -> List[Tuple[str, int, bool]]:
        return list(self._columns)

    def dump_column(self, idx: int, max_rows_per_world: int = 256):
        name, elem_bytes, is_float = self._columns[idx]
        cap = self.num_worlds * max_rows_per_world * elem_bytes
        buf = np.empty(cap, dtype=np.uint8)
        counts = np.zeros(self.num_worlds, dtype=np.int32)
        n = self.lib.sim_column_dump(
            self.handle, idx, buf.ctypes.data, buf.nbytes,
            counts.ctypes.data_as(C.POINTER(C.c_int32)))
        if n < 0:
            detail = ""
            if self.backend == "hip":
                detail = ": " + runtime_lib().mwhip_last_error().decode()
            raise RuntimeError(f"sim_column_dump({name}) -> {n}{detail}")
        return buf[: n * elem_bytes].reshape(n, elem_bytes).copy(), counts

    def dump_all(self, max_rows_per_world: int = 256):
        return {self._columns[i][0]: self.dump_column(i, max_rows_per_world)
                for i in range(len(self._columns))}

    def run_taskgraph(self, taskgraph_id: int) -> None:
        """Replays ONE task graph of the simulator (HIP backend; test probes)."""
        rc = self.lib.sim_hip_run_taskgraph(self.handle, taskgraph_id)
        if rc != 0:
            raise RuntimeError(f"sim_hip_run_taskgraph({taskgraph_id}) -> {rc}")

    def render(self) -> None:
        """MWCudaExecutor::buildRenderGraph + run: TLAS build and ray cast of every
        view into the simulator's "rgb" / "depth" tensors (HIP backend)."""
        rc = self.lib.sim_hip_render(self.handle)
        if rc != 0:
            raise RuntimeError(f"sim_hip_render -> {rc}: "
                               f"{runtime_lib().mwhip_last_error().decode()}")

    def render_graph(self) -> int:
        """Launch-graph handle of the render pass (for step_async(graph=...))."""
        return int(self.lib.sim_hip_render_graph(self.handle))

    def dump_column_raw(self, idx: int, max_rows: int):
        """Column `idx` in table order, destroyed rows included (HIP backend)."""
        name, elem_bytes, _ = self._columns[idx]
        buf = np.empty(max_rows * elem_bytes, dtype=np.uint8)
        n = self.lib.sim_column_dump_raw(self.handle, idx, buf.ctypes.data, buf.nbytes)
        if n < 0:
            raise RuntimeError(f"sim_column_dump_raw({name}) -> {n}")
        return buf[: n * elem_bytes].reshape(n, elem_bytes).copy()

    def hip_exec(self) -> int:
        return int(self.lib.sim_hip_exec(self.handle) or 0)

    # ---- stream-ordered stepping (HIP backend) -----------------------------------
    def stream(self) -> int:
        """hipStream_t of the executor's private stream (mwhip_stream)."""
        return int(runtime_lib().mwhip_stream(self.hip_exec()) or 0)

    def _pack_descriptor(self, names: List[str]):
        n = len(names)
        ptrs = (C.c_void_p * n)(*[self.tensor_ptr(name) for name in names])
        words = (C.c_uint32 * n)()
        for i, name in enumerate(names):
            _, dtype, dims, _ = self._tensor_info[name]
            if np.dtype(dtype).itemsize != 4:
                raise TypeError(f"{name}: only 4-byte element types are packed")
            words[i] = int(np.prod(dims[1:])) if len(dims) > 1 else 1
        return n, ptrs, words

    def packed_step_graph(self, names: List[str], dst_ptr: int) -> int:
        """A copy of the step graph whose last node packs the exported tensors
        `names` into one [worlds, words] int32 record per world at device
        address `dst_ptr` (mwhip_build_launch_graph_with_pack); replay it with
        step_async(graph=...)."""
        rt = runtime_lib()
        n, ptrs, words = self._pack_descriptor(names)
        out = C.c_uint64(0)
        rc = rt.mwhip_build_launch_graph_with_pack(
            self.hip_exec(), self.lib.sim_hip_step_graph(self.handle), n, ptrs,
            words, self.num_worlds, dst_ptr, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"mwhip_build_launch_graph_with_pack -> {rc}: "
                               f"{rt.mwhip_last_error().decode()}")
        return int(out.value)

    def step_async(self, n: int = 1, graph: int = 0) -> None:
        """Queues n replays of the step graph on the executor's stream without
        waiting for them (MWCudaExecutor::runAsync, reference mw_gpu.hpp:146):
        work that consumes the exported tensors must be ordered after this
        stream (events / sync()), as with any stream-ordered producer."""
        if self._async is None:     # (looked up once: this is a per-step call)
            rt = runtime_lib()
            exec_ = self.hip_exec()
            self._async = (rt, exec_, self.lib.sim_hip_step_graph(self.handle),
                           rt.mwhip_stream(exec_))
        rt, exec_, step_graph, stream = self._async
        graph = graph or step_graph
        for _ in range(n):
            rc = rt.mwhip_run_async(exec_, graph, stream)
            if rc != 0:
                raise RuntimeError(
                    f"mwhip_run_async -> {rc}: {rt.mwhip_last_error().decode()}")

    def set_input_ring(self, name: str, ring_ptr: int, num_slots: int) -> None:
        """The k-th replay after this call starts by copying slot k % num_slots
        of the device-resident ring at `ring_ptr` (num_slots x the tensor's
        bytes) into exported tensor `name` (mwhip_set_input_ring); ring_ptr = 0
        removes the ring."""
        rt = runtime_lib()
        _, dtype, dims, _ = self._tensor_info[name]
        slot_bytes = int(np.prod(dims)) * dtype.itemsize
        rc = rt.mwhip_set_input_ring(self.hip_exec(), self.tensor_ptr(name),
                                     ring_ptr or None, slot_bytes, num_slots)
        if rc != 0:
            raise RuntimeError(f"mwhip_set_input_ring -> {rc}: "
                               f"{rt.mwhip_last_error().decode()}")

    def stream_wait_replays(self, hip_stream: int) -> None:
        """Makes `hip_stream` (a hipStream_t) wait for every replay queued so
        far, without putting anything on the executor's stream
        (mwhip_stream_wait_replays)."""
        rt = runtime_lib() if self._async is None else self._async[0]
        rc = rt.mwhip_stream_wait_replays(self.hip_exec(), hip_stream)
        if rc != 0:
            raise RuntimeError(f"mwhip_stream_wait_replays -> {rc}: "
                               f"{rt.mwhip_last_error().decode()}")

    def pack_rows_async(self, names: List[str], dst_ptr: int) -> None:
        """Queues, behind the replays queued so far, the packing of the exported
        tensors `names` into one [worlds, words] int32 record per world at
        device address `dst_ptr` (mwhip_pack_rows)."""
        rt = runtime_lib()
        n, ptrs, words = self._pack_descriptor(names)
        rc = rt.mwhip_pack_rows(self.hip_exec(), n, ptrs, words,
                                self.num_worlds, dst_ptr)
        if rc != 0:
            raise RuntimeError(
                f"mwhip_pack_rows -> {rc}: {rt.mwhip_last_error().decode()}")

    def sync(self) -> None:
        """Waits for the queued replays; raises on a device-side error flag."""
        rt = runtime_lib()
        rc = rt.mwhip_synchronize(self.hip_exec())
        if rc != 0:
            raise RuntimeError(
                f"mwhip_synchronize -> {rc}: {rt.mwhip_last_error().decode()}")

    def profile(self, reps: int = 20, graph: int = 0):
        """Per-kernel timing of one step (HIP events on the executor's stream,
        kernels queued back to back behind a gate).  Advances the simulation by
        `reps` steps.  graph: another launch graph of this executor (e.g.
        render_graph()) instead of the step graph.  Returns a list of dicts."""
        rt = runtime_lib()
        stats = (KernelStat * 512)()
        n = rt.mwhip_profile(self.hip_exec(),
                             graph or self.lib.sim_hip_step_graph(self.handle),
                             reps, stats, 512)
        if n < 0:
            raise RuntimeError(f"mwhip_profile -> {n}: {rt.mwhip_last_error().decode()}")
        return [dict(name=stats[i].name.decode(), kind=int(stats[i].node_kind),
                     avg_us=float(stats[i].avg_us), algo_bytes=float(stats[i].algo_bytes),
                     rows=float(stats[i].rows),
                     io_declared=bool(stats[i].io_declared),
                     workgroups=int(stats[i].workgroups),
                     node_index=int(stats[i].node_index)) for i in range(n)]
