"""CPU-only, world_size 2 over gloo: the multi-GPU path of
madrona_amd.distributed (shard by global world index + all-gather of the
observation tensors).  The per-rank executor here is the reference CPU backend
(test infrastructure) so the test runs without a GPU; on the MI355X node the
same code drives the HIP backend over RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from madrona_amd.distributed import ShardedSimulator, shard_for
from madrona_amd.simlib import Simulator, ref_lib_path

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


CASES = {
    # sim: (gathered tensors, total worlds, steps)
    "cartpole": (["state", "reward", "done"], 64, 50),
    # BASELINE config 4's shape: the observation columns of the Hide-and-Seek
    # worlds (2.9 KB per world) packed into one all-gather per step
    "hideseek": (["self_obs", "agent_obs", "box_obs", "ramp_obs", "lidar",
                  "reward", "done"], 8, 12),
}


def _worker(rank, world_size, port, sim_name, out_dir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # (gloo otherwise picks its interface by resolving the host name, which in
    # this container may not resolve -- or resolve slowly: the loopback device)
    if os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    # (a rendezvous that cannot complete -- the port taken between _free_port()
    # and here -- fails after a minute instead of gloo's default half hour)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world_size,
                            timeout=datetime.timedelta(seconds=45))
    # (the parent retries a run only while no rank got this far)
    open(os.path.join(out_dir, f"rendezvous_ok_{rank}"), "w").close()
    try:
        names, total_worlds, steps = CASES[sim_name]
        shard = shard_for(rank, world_size, total_worlds=total_worlds)

        def make_sim(num_worlds, world_base):
            return Simulator(ref_lib_path(sim_name), num_worlds, seed=5,
                             num_workers=1, world_base=world_base)

        sharded = ShardedSimulator(make_sim, shard, names)
        gathered = None
        for _ in range(steps):
            gathered = sharded.step(1)
        if rank == 0:
            np.savez(os.path.join(out_dir, "gathered.npz"),
                     **{k: v.numpy() for k, v in gathered.items()})
        sharded.close()
    finally:
        dist.destroy_process_group()


def test_shard_arithmetic():
    s = shard_for(3, 8, total_worlds=65536)
    assert (s.worlds_per_rank, s.world_base, s.total_worlds) == (8192, 24576, 65536)
    s = shard_for(1, 2, worlds_per_rank=4096)
    assert (s.world_base, s.total_worlds) == (4096, 8192)
    with pytest.raises(ValueError):
        shard_for(0, 3, total_worlds=64)


@pytest.mark.parametrize("sim_name", sorted(CASES))
def test_two_rank_allgather_is_partition_invariant(built, tmp_path, sim_name):
    if not os.path.exists(ref_lib_path(sim_name)):
        pytest.skip("oracle/_ref not built here (no /root/reference)")
    names, total_worlds, steps = CASES[sim_name]
    import glob
    import time
    for attempt in range(3):
        ctx = mp.spawn(_worker, args=(2, _free_port(), sim_name, str(tmp_path)),
                       nprocs=2, join=False)
        error = None
        deadline = time.monotonic() + 150
        try:
            while not ctx.join(timeout=5):
                if time.monotonic() > deadline:
                    raise TimeoutError("the two ranks did not finish in 150 s")
            break
        except Exception as e:      # noqa: BLE001
            error = e
            for proc in ctx.processes:      # (exactly the processes started here)
                if proc.is_alive():
                    proc.kill()
        # Only a lost rendezvous (the port taken between _free_port() and
        # init_process_group) is retried, on a new port: once a rank got past
        # it, a failure is a failure of what is under test.
        if glob.glob(os.path.join(str(tmp_path), "rendezvous_ok_*")) or attempt == 2:
            raise error
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))

    # one process owning all worlds must produce the same tensors bit for bit
    with Simulator(ref_lib_path(sim_name), total_worlds, seed=5, num_workers=1) as s:
        s.step(steps)
        for name in names:
            assert np.array_equal(got[name].view(np.uint8),
                                  s.read_tensor(name).view(np.uint8)), name
