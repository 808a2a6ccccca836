"""The Python bridge beyond Tensor: TrainInterface, Tensor
from a torch tensor, and the XLA custom-call entry points of
madrona::py::JAXInterface (include/madrona/py/bindings.hpp; reference
include/madrona/py/utils.hpp:143-204, bindings.inl:102-121,
src/python/utils.cpp:403-560).  jax is not installed here: the entry points are
called through their capsules the way XLA calls a registered target (CPU:
fn(out, in) with in[0] = the simulator's address, in[1] = the token; GPU:
fn(stream, buffers, opaque, len) with opaque = the address), on the module's
Manager-shaped demo class."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from madrona_amd.simlib import HIP_BUILD_DIR


@pytest.fixture(scope="module")
def pymod(built):
    import torch  # noqa: F401  (torch's HIP runtime first, see simlib)
    if HIP_BUILD_DIR not in sys.path:
        sys.path.insert(0, HIP_BUILD_DIR)
    import _madrona_amd_py
    return _madrona_amd_py


def test_train_interface_pytrees(pymod):
    sim = pymod._DemoTrainSim(5)
    iface = sim.train_interface()
    assert iface.step_inputs() == {
        "actions": {"move": ((5, 2), "int32")}, "resets": ((5,), "int32"),
        "sim_ctrl": ((1,), "int32"), "pbt": {}}
    assert iface.step_outputs() == {
        "obs": {"self": ((5, 3), "float32")}, "rewards": ((5,), "float32"),
        "dones": ((5,), "int32"), "stats": {}, "pbt": {}}


def test_tensor_from_torch_is_a_view(pymod):
    import torch
    src = torch.arange(12, dtype=torch.int16).reshape(3, 4)
    t = pymod.Tensor(src)
    assert t.shape == [3, 4] and t.type == pymod.TensorElementType.Int16
    assert not t.is_on_gpu and t.device_ptr == src.data_ptr()
    back = t.to_torch()
    back[2, 3] = -5
    assert int(src[2, 3]) == -5
    with pytest.raises(ValueError):
        pymod.Tensor(torch.zeros(4, 4).t())         # not contiguous
    with pytest.raises(TypeError):
        pymod.Tensor(torch.zeros(4, dtype=torch.float64))


def test_cpu_custom_call_entry_points(pymod):
    """init then two steps through the capsules, buffers laid out as
    jax_register.py's lowering passes them."""
    n = 4
    sim = pymod._DemoTrainSim(n)
    entry = sim.jax(False, register=False)
    assert entry["platform"] == "cpu"
    assert entry["step_inputs_iface"]["actions"]["move"] == ((n, 2), "int32")
    sim_addr = np.array([entry["sim_ptr"]], dtype=np.uint64)
    assert entry["sim_encode"] == sim_addr.tobytes()
    token = np.empty((0,), np.float32)

    obs = np.full((n, 3), -1, np.float32)
    out_token = np.empty((0,), np.float32)
    pymod._call_cpu_custom_call(entry["init"], [obs.ctypes.data, out_token.ctypes.data],
                                [sim_addr.ctypes.data, token.ctypes.data])
    assert np.array_equal(obs, np.zeros((n, 3), np.float32))

    rewards = np.zeros(n, np.float32)
    dones = np.zeros(n, np.int32)
    for step in (1, 2):
        actions = (np.arange(n * 2, dtype=np.int32).reshape(n, 2) + step)
        resets = np.array([0, 1, 0, 1], np.int32)
        ctrl = np.zeros(1, np.int32)
        pymod._call_cpu_custom_call(
            entry["step"],
            [obs.ctypes.data, rewards.ctypes.data, dones.ctypes.data,
             out_token.ctypes.data],
            [sim_addr.ctypes.data, token.ctypes.data, actions.ctypes.data,
             resets.ctypes.data, ctrl.ctypes.data])
        assert sim.steps == step
        assert np.array_equal(obs[:, 0], actions.sum(1).astype(np.float32))
        assert np.array_equal(obs[:, 1], resets.astype(np.float32))
        assert np.all(obs[:, 2] == step)
        assert np.array_equal(rewards, obs[:, 0] / 2) and np.array_equal(dones, resets)
    # the simulator's own tensors hold what was copied in / out
    assert np.array_equal(sim.action_tensor().to_torch().numpy(), actions)
    assert np.array_equal(sim.obs_tensor().to_torch().numpy(), obs)


def test_jax_registration_needs_jax(pymod):
    """setup() goes through madrona_amd/jax_register.py, which says what is
    missing instead of failing somewhere inside XLA."""
    try:
        import jax  # noqa: F401
        pytest.skip("jax is installed: covered by the jax-side test instead")
    except ImportError:
        pass
    sim = pymod._DemoTrainSim(2)
    with pytest.raises(ImportError, match="needs jax"):
        sim.jax(False)


@pytest.mark.gpu
def test_gpu_custom_call_entry_points(pymod):
    """The GPU flavour: device buffers, a caller's stream, the opaque."""
    import torch
    n = 6
    sim = pymod._DemoTrainSim(n, 0)
    entry = sim.jax(True, register=False)
    assert entry["platform"] == "gpu"
    stream = torch.cuda.Stream()
    dev = torch.device("cuda", 0)
    first = torch.zeros(1, dtype=torch.int64, device=dev)    # (operand 0: unused on GPU)
    token = torch.empty(0, dtype=torch.float32, device=dev)
    obs = torch.full((n, 3), -1.0, device=dev)
    with torch.cuda.stream(stream):
        pymod._call_gpu_custom_call(
            entry["init"], stream.cuda_stream,
            [first.data_ptr(), token.data_ptr(), obs.data_ptr(), token.data_ptr()],
            entry["sim_encode"])
    stream.synchronize()
    assert torch.equal(obs.cpu(), torch.zeros(n, 3))

    actions = torch.arange(n * 2, dtype=torch.int32, device=dev).reshape(n, 2)
    resets = (torch.arange(n, device=dev) % 2).to(torch.int32)
    ctrl = torch.zeros(1, dtype=torch.int32, device=dev)
    rewards = torch.zeros(n, device=dev)
    dones = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        pymod._call_gpu_custom_call(
            entry["step"], stream.cuda_stream,
            [first.data_ptr(), token.data_ptr(), actions.data_ptr(),
             resets.data_ptr(), ctrl.data_ptr(), obs.data_ptr(), rewards.data_ptr(),
             dones.data_ptr(), token.data_ptr()],
            entry["sim_encode"])
    stream.synchronize()
    assert torch.equal(obs[:, 0].cpu(), actions.sum(1).float().cpu())
    assert torch.equal(obs[:, 1].cpu(), resets.float().cpu())
    assert torch.all(obs[:, 2] == 1)
    assert torch.equal(rewards.cpu(), obs[:, 0].cpu() / 2)
    assert torch.equal(dones.cpu(), resets.cpu())
    # the simulator's exported tensor is a live PyTorch-ROCm tensor
    t = sim.obs_tensor().to_torch()
    assert t.is_cuda and torch.equal(t, obs)


def test_step_buffers_with_stats_and_pbt(pymod):
    """A simulator that exports a stats tensor and pbt tensors both ways: the
    result list jax_register.step_buffer_specs declares is exactly what
    cpuCopyStepOutputs writes, in its order (observations..., rewards, dones,
    stats..., pbt...) -- nothing lands in a neighbour's buffer, nothing is
    written behind the last one."""
    from madrona_amd import jax_register
    n = 5
    sim = pymod._DemoTrainSim(n, -1, True)
    entry = sim.jax(False, register=False)
    ins, outs = entry["step_inputs_iface"], entry["step_outputs_iface"]
    assert ins["pbt"] == {"policy": ((n,), "int32")}
    assert outs["stats"] == {"episodes": ((n,), "int32")}
    assert outs["pbt"] == {"fitness": ((n,), "float32")}
    names, specs = jax_register.step_buffer_specs(outs)
    assert names == {"obs": ["self"], "stats": ["episodes"], "pbt": ["fitness"]}
    assert specs == [((n, 3), "float32"), ((n,), "float32"), ((n,), "int32"),
                     ((n,), "int32"), ((n,), "float32")]
    assert jax_register.step_input_order(ins) == (["move"], ["policy"])
    assert jax_register.xla_platforms("gpu") == ["ROCM", "gpu"]
    assert jax_register.xla_platforms("cpu") == ["cpu"]

    # result buffers allocated from the declared specs, each with a guard word
    # behind it, and a guard entry behind the token in the pointer list
    bufs = [np.full(int(np.prod(shape)) + 1, -7, np.dtype(dt)) for shape, dt in specs]
    token = np.empty((0,), np.float32)
    sim_addr = np.array([entry["sim_ptr"]], dtype=np.uint64)
    actions = np.arange(n * 2, dtype=np.int32).reshape(n, 2)
    resets = np.array([1, 0, 0, 1, 0], np.int32)
    ctrl = np.zeros(1, np.int32)
    policy = np.arange(n, dtype=np.int32) * 3
    pymod._call_cpu_custom_call(
        entry["step"], [b.ctypes.data for b in bufs] + [token.ctypes.data],
        [sim_addr.ctypes.data, token.ctypes.data, actions.ctypes.data,
         resets.ctypes.data, ctrl.ctypes.data, policy.ctypes.data])
    obs, rewards, dones, episodes, fitness = bufs
    assert all(b[-1] == -7 for b in bufs)                      # guards intact
    assert np.array_equal(obs[:-1].reshape(n, 3)[:, 0], actions.sum(1))
    assert np.array_equal(rewards[:-1], actions.sum(1) / 2)
    assert np.array_equal(dones[:-1], resets)
    assert np.array_equal(episodes[:-1], 10 + np.arange(n))
    assert np.array_equal(fitness[:-1], (policy - np.arange(n)).astype(np.float32))
