"""API conformance of the header overlay: the symbols a real
Madrona simulator / Manager / Python binding names exist, compile for gfx950 and
the host (madrona_amd/_build/libapi_conformance.so is built from
tests/shims/api_conformance{.hip,_host.cpp}), and behave.

CPU part: host containers (DynArray / HeapArray / InlineArray), host atomics,
py::Tensor, the pybind11 module's zero-copy DLPack export on host memory.
GPU part (-m gpu): the conformance simulator runs through the executor shim --
device atomics, SpinLock, TmpAllocator, GPUImplConsts, HostPrint."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from madrona_amd.simlib import HIP_BUILD_DIR, Simulator, hip_lib_path

LIB = os.path.join(HIP_BUILD_DIR, "libapi_conformance.so")
PYMOD_DIR = HIP_BUILD_DIR


@pytest.fixture(scope="module")
def conf(built):
    import torch  # noqa: F401  (torch's HIP runtime first, see simlib)
    C.CDLL(os.path.join(HIP_BUILD_DIR, "libmadrona_hip.so"), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB)
    lib.conf_containers.restype = C.c_int
    lib.conf_tensor_bytes.restype = C.c_int64
    lib.conf_tensor_bytes.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64),
                                      C.c_int32, C.c_int32]
    lib.conf_run.restype = C.c_int
    lib.conf_run.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    return lib


@pytest.fixture(scope="module")
def pymod(built):
    if PYMOD_DIR not in sys.path:
        sys.path.insert(0, PYMOD_DIR)
    import _madrona_amd_py
    return _madrona_amd_py


def test_host_containers_and_atomics(conf):
    assert conf.conf_containers() == 0


def test_row_chain_on_the_host(conf):
    """madrona::mwhip::rowChain (taskgraph.inl): two systems over
    one row -- each gets the components its signature names, the first runs
    first -- and the name profiles give the node."""
    name = C.create_string_buffer(256)
    conf.conf_row_chain.argtypes = [C.c_char_p, C.c_int32]
    assert conf.conf_row_chain(name, 256) == 0
    assert name.value.decode() == \
        "chain[conformance_chain::openSystem > conformance_chain::moveSystem]"


def test_host_tracing_log(conf, tmp_path, monkeypatch):
    """madrona/tracing.hpp (reference include/madrona/tracing.hpp,
    src/common/tracing.cpp:44-58): N event codes, then N time stamps, int64."""
    monkeypatch.setenv("MADRONA_MWGPU_TRACE_NAME", "conf")
    conf.conf_tracing.argtypes = [C.c_char_p]
    assert conf.conf_tracing((str(tmp_path) + "/").encode()) == 0
    log = np.fromfile(tmp_path / "conf_madrona_host_tracing.bin", dtype=np.int64)
    assert log.shape == (6,) and log[:3].tolist() == [0, 1, 2]
    assert log[3] <= log[4] <= log[5]


def test_py_tensor_reference_constructor(conf):
    buf = np.zeros((4, 3, 2), dtype=np.float32)
    dims = (C.c_int64 * 3)(4, 3, 2)
    # TensorElementType::Float32 == 6 (reference py/utils.hpp:58-66)
    assert conf.conf_tensor_bytes(buf.ctypes.data, 6, dims, 3, -1) == buf.nbytes
    assert conf.conf_tensor_bytes(buf.ctypes.data, 6, dims, 3, 0) == buf.nbytes
    assert conf.conf_tensor_bytes(buf.ctypes.data, 2, dims, 3, -1) == 4 * 3 * 2 * 2


def test_pybind_tensor_to_torch_host_zero_copy(pymod):
    import torch
    buf = np.arange(24, dtype=np.int32).reshape(2, 3, 4)
    t = pymod.Tensor(buf.ctypes.data, pymod.TensorElementType.Int32, [2, 3, 4])
    assert not t.is_on_gpu and t.shape == [2, 3, 4] and t.bytes_per_item == 4
    assert t.__dlpack_device__() == (1, 0)
    tt = t.to_torch()
    assert tt.dtype == torch.int32 and tuple(tt.shape) == (2, 3, 4)
    assert tt.data_ptr() == buf.ctypes.data          # a view, not a copy
    buf[1, 2, 3] = -7
    assert int(tt[1, 2, 3]) == -7
    tt[0, 0, 0] = 99
    assert buf[0, 0, 0] == 99
    # torch.from_dlpack goes through __dlpack__ / __dlpack_device__
    assert torch.from_dlpack(t).data_ptr() == buf.ctypes.data
    for et, dt in [(pymod.TensorElementType.UInt8, torch.uint8),
                   (pymod.TensorElementType.Int8, torch.int8),
                   (pymod.TensorElementType.Int16, torch.int16),
                   (pymod.TensorElementType.Int64, torch.int64),
                   (pymod.TensorElementType.Float16, torch.float16),
                   (pymod.TensorElementType.Float32, torch.float32)]:
        raw = np.zeros(64, dtype=np.uint8)
        assert pymod.Tensor(raw.ctypes.data, et, [4]).to_torch().dtype == dt


@pytest.mark.gpu
def test_conformance_simulator_on_device(conf, capfd):
    W, steps = 70, 5
    out = np.zeros((W, 4), dtype=np.uint32)
    assert conf.conf_run(W, steps, out.ctypes.data) == 0
    hits, sums = out[:, 0], out[:, 1].view(np.float32)
    assert (hits == steps).all()                    # AtomicU32Ref::fetch_add
    assert (sums == 0.5 * steps).all()              # AtomicFloatRef::fetch_add
    assert (out[:, 2] == 0).all()                   # CAS acquire + release store
    # tmpAlloc'd word (2) + roundUpAlloc(1)/256 (1) + InlineArray size (2)
    assert (out[:, 3] == 5).all(), out[:4]
    text = capfd.readouterr().out
    assert "conformance: world 0 hits 1 sum 0.500000" in text


@pytest.mark.gpu
def test_host_print_from_a_simulator(built, capfd):
    """mwGPU::HostPrint::log from a system: every argument type, exactly once,
    also when the replays are queued without being waited for one by one."""
    with Simulator(hip_lib_path("sort_stress"), 8, seed=7) as s:
        s.step_async(6)
        s.sync()
    text = capfd.readouterr().out
    lines = [ln for ln in text.splitlines() if ln.startswith("sort_stress: world 3")]
    assert len(lines) == 1, text
    assert " step 3 items " in lines[0] and "key 20015998343868 ptr 0x" in lines[0]


@pytest.mark.gpu
def test_host_print_ring_overflow(built, capfd):
    """More messages in one replay than the ring holds (4096 worlds speak at
    once, 1024 records): what fits is printed once each, the rest is counted as
    dropped -- a message that does not fit takes no ticket, so the host never
    has to guess whether a missing record is late or lost -- and the channel
    works afterwards (world 3's message of the next step)."""
    worlds = 4096
    with Simulator(hip_lib_path("sort_stress"), worlds, seed=7, flags=4) as s:
        s.step(2)
        first = capfd.readouterr().out
        s.step(1)
        second = capfd.readouterr().out
    chatty = [ln for ln in first.splitlines()
              if ln.startswith("sort_stress: chatty world ")]
    ids = [int(ln.rsplit(" ", 1)[1]) for ln in chatty]
    assert len(set(ids)) == len(ids) and all(0 <= i < worlds for i in ids)
    dropped = [ln for ln in first.splitlines() if "HostPrint ring overflow" in ln]
    n_dropped = sum(int(ln.split("overflow, ")[1].split(" ")[0]) for ln in dropped)
    assert len(ids) + n_dropped == worlds, (len(ids), n_dropped)
    assert len(ids) >= 1024
    assert sum(ln.startswith("sort_stress: world 3 step 3") for ln in
               second.splitlines()) == 1, second[-2000:]


@pytest.mark.gpu
def test_pybind_tensor_to_torch_device(pymod, built):
    """The C++ py::Tensor route and the ctypes route give the same live view of
    an exported column (device kDLROCM -> torch 'cuda')."""
    import torch
    from madrona_amd.tensor import to_torch
    with Simulator(hip_lib_path("cartpole"), 64) as s:
        s.step(3)
        name = s.tensor_names[0]
        _, dtype, dims, _ = s.tensor_meta(name)
        et = {np.dtype(np.float32): pymod.TensorElementType.Float32,
              np.dtype(np.int32): pymod.TensorElementType.Int32}[np.dtype(dtype)]
        t = pymod.Tensor(s.tensor_ptr(name), et, list(dims), 0)
        assert t.is_on_gpu and t.__dlpack_device__() == (10, 0)
        a = t.to_torch()
        b = to_torch(s, name)
        assert a.is_cuda and a.data_ptr() == b.data_ptr() == s.tensor_ptr(name)
        assert torch.equal(a, b)
        s.step(1)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
