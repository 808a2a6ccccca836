"""GPU, world_size 2 on ONE MI355X: the multi-GPU code path of the HIP backend
-- ShardedSimulator with per-rank executors, the pack node inside the step
graph, stream-ordered hand-over to the collective, double-buffered send
records -- run by two processes that share device 0.  RCCL refuses two ranks on
one device, so the collective itself goes over gloo through pinned host memory
(madrona_amd/distributed.py:_exchange); everything else is what `bench.py
--gpus N` runs over RCCL.  Checked: the gathered observation tensors of the
2 x W/2 run equal, bit for bit, those of one executor owning all W worlds."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NAMES = ["self_obs", "agent_obs", "box_obs", "ramp_obs", "lidar", "reward", "done"]
TOTAL_WORLDS, STEPS = 512, 40


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _actions(step, worlds, base):
    rng = np.random.default_rng(4000 + step)
    shape = (TOTAL_WORLDS, 5)
    a = np.stack([rng.integers(0, 4, shape), rng.integers(0, 8, shape),
                  rng.integers(-2, 3, shape), rng.integers(0, 2, shape)],
                 -1).astype(np.int32)
    return a[base:base + worlds]


def _worker(rank, world_size, port, out_dir, use_ring):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch
    import torch.distributed as dist
    from madrona_amd.distributed import ShardedSimulator, shard_for
    from madrona_amd.simlib import Simulator, hip_lib_path

    torch.cuda.set_device(0)
    import datetime
    # (gloo otherwise resolves the host name to pick its interface)
    if os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world_size,
                            timeout=datetime.timedelta(seconds=90))
    try:
        shard = shard_for(rank, world_size, total_worlds=TOTAL_WORLDS)

        def make_sim(num_worlds, world_base):
            return Simulator(hip_lib_path("hideseek"), num_worlds, seed=5,
                             gpu_id=0, world_base=world_base, flags=40)

        sharded = ShardedSimulator(make_sim, shard, NAMES)
        gathered = None
        ring = None
        if use_ring:
            # the rank's shard of every step's actions, resident on the device:
            # replay k of the (pack-carrying) step graphs copies slot k itself
            # (mwhip_set_input_ring rebuilds those graphs, pack node included)
            ring = torch.from_numpy(np.stack([
                _actions(step, shard.worlds_per_rank, shard.world_base)
                for step in range(1, STEPS + 1)])).cuda()
            sharded.sim.set_input_ring("action", ring.data_ptr(), STEPS)
        t0 = time.perf_counter()
        for step in range(1, STEPS + 1):
            if not use_ring:
                sharded.sim.write_tensor(
                    "action", _actions(step, shard.worlds_per_rank, shard.world_base))
            gathered = sharded.step(1)
        torch.cuda.synchronize()
        sharded.sync()
        ms = (time.perf_counter() - t0) / STEPS * 1e3
        np.savez(os.path.join(out_dir, f"gathered_{rank}.npz"), ms_per_step=ms,
                 **{k: v.cpu().numpy() for k, v in gathered.items()})
        sharded.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_ring", [False, True])
def test_two_ranks_share_one_gpu(built, tmp_path, use_ring):
    import torch
    import torch.multiprocessing as mp
    from madrona_amd.simlib import Simulator, hip_lib_path

    assert torch.cuda.is_available()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), use_ring), nprocs=2, join=True)
    got = [np.load(os.path.join(str(tmp_path), f"gathered_{r}.npz")) for r in (0, 1)]

    with Simulator(hip_lib_path("hideseek"), TOTAL_WORLDS, seed=5, flags=40) as s:
        for step in range(1, STEPS + 1):
            s.write_tensor("action", _actions(step, TOTAL_WORLDS, 0))
            s.step(1)
        for name in NAMES:
            want = s.read_tensor(name).view(np.uint8)
            for r in (0, 1):        # every rank holds the full gathered tensor
                assert np.array_equal(got[r][name].view(np.uint8), want), (name, r)
    print("per-rank ms/step:", [float(g["ms_per_step"]) for g in got])
