"""Multi-GPU execution: worlds shard trivially (no cross-world reads anywhere in
the engine), so the job is N independent executors -- one process
per GPU -- each owning a contiguous range of *global* world indices, plus ONE
exchange per step: an all-gather of the exported observation tensor over
RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in CPU tests).

World RNG keys derive from the global world index (``world_base`` of the
simulator C API), so results do not depend on how worlds are partitioned.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional


@dataclass
class Shard:
    rank: int
    world_size: int
    worlds_per_rank: int

    @property
    def world_base(self) -> int:
        return self.rank * self.worlds_per_rank

    @property
    def total_worlds(self) -> int:
        return self.world_size * self.worlds_per_rank


def shard_for(rank: int, world_size: int, total_worlds: Optional[int] = None,
              worlds_per_rank: Optional[int] = None) -> Shard:
    """Either weak scaling (fixed ``worlds_per_rank``) or an even split of
    ``total_worlds`` (must divide)."""
    if worlds_per_rank is None:
        if total_worlds is None or total_worlds % world_size != 0:
            raise ValueError("total_worlds must be a multiple of world_size")
        worlds_per_rank = total_worlds // world_size
    return Shard(rank, world_size, worlds_per_rank)


class ShardedSimulator:
    """One rank's executor + the per-step observation all-gather."""

    def __init__(self, make_sim, shard: Shard, obs_names: List[str], group=None):
        """``make_sim(num_worlds, world_base)`` -> simulator with
        ``step()``/tensor access (madrona_amd.simlib.Simulator or compatible);
        ``obs_names`` are gathered after every step."""
        import torch.distributed as dist

        self.shard = shard
        self.sim = make_sim(shard.worlds_per_rank, shard.world_base)
        self.obs_names = list(obs_names)
        self.group = group
        self._dist = dist
        self._local: Dict[str, "torch.Tensor"] = {}
        self._global: Dict[str, "torch.Tensor"] = {}

    def _local_tensor(self, name: str):
        if name not in self._local:
            from .tensor import to_torch
            self._local[name] = to_torch(self.sim, name)
        return self._local[name]

    def step(self, n: int = 1):
        """Steps the local worlds, then gathers every observation tensor:
        out[name] has shape [total_worlds, ...]."""
        import torch

        self.sim.step(n)
        for name in self.obs_names:
            local = self._local_tensor(name)
            if name not in self._global:
                shape = (self.shard.total_worlds,) + tuple(local.shape[1:])
                self._global[name] = torch.empty(shape, dtype=local.dtype,
                                                 device=local.device)
            if self.shard.world_size == 1:
                self._global[name].copy_(local)
            else:
                self._dist.all_gather_into_tensor(
                    self._global[name], local.contiguous(), group=self.group)
        return self._global

    def close(self):
        self.sim.close()
