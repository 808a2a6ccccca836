"""Multi-GPU execution: worlds shard trivially (no cross-world reads anywhere in
the engine), so the job is N independent executors -- one process
per GPU -- each owning a contiguous range of *global* world indices, plus ONE
exchange per step: a single all-gather of the packed observation tensors over
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
    """One rank's executor + the per-step observation all-gather.

    The exported observation columns are separate buffers; gathering them one by
    one would cost one collective launch (tens of microseconds over xGMI) per
    tensor per step, more than the whole simulation step.  They are therefore
    packed into ONE [worlds, words] int32 record per world with a single
    ``torch.cat`` and exchanged with ONE ``all_gather_into_tensor``; the
    per-tensor results are strided views of the gathered buffer."""

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
        self._local_words = None    # per tensor: [W, n_i] int32 view
        self._packed_local = None
        self._packed_global = None
        self._global: Dict[str, "torch.Tensor"] = {}

    def _setup(self):
        import torch
        from .tensor import to_torch

        W = self.shard.worlds_per_rank
        self._local_words = []
        layout = []
        offset = 0
        for name in self.obs_names:
            local = to_torch(self.sim, name)
            if local.element_size() != 4:
                raise TypeError(f"{name}: only 4-byte element types are packed")
            words = local.view(torch.int32).reshape(W, -1)
            self._local_words.append(words)
            layout.append((name, local.dtype, tuple(local.shape[1:]), offset,
                           words.shape[1]))
            offset += words.shape[1]

        device = self._local_words[0].device if self._local_words else "cpu"
        self._packed_local = torch.empty((W, offset), dtype=torch.int32,
                                         device=device)
        self._packed_global = torch.empty((self.shard.total_worlds, offset),
                                          dtype=torch.int32, device=device)
        for name, dtype, tail, off, n in layout:
            view = self._packed_global[:, off:off + n].view(dtype)
            self._global[name] = view.unflatten(1, tail) if tail else view[:, 0]

    def step(self, n: int = 1):
        """Steps the local worlds, then gathers the observation tensors:
        out[name] has shape [total_worlds, ...]."""
        import torch

        self.sim.step(n)
        if not self.obs_names:
            return self._global
        if self._local_words is None:
            self._setup()

        torch.cat(self._local_words, dim=1, out=self._packed_local)
        if self._packed_local.is_cuda:
            # the executor steps on its own stream: the pack must have read the
            # exported columns before the next step may overwrite them (the
            # collective itself then overlaps with that step, it only reads the
            # packed copy)
            torch.cuda.current_stream(self._packed_local.device).synchronize()
        if self.shard.world_size == 1:
            self._packed_global.copy_(self._packed_local)
        else:
            self._dist.all_gather_into_tensor(
                self._packed_global, self._packed_local, group=self.group)
        return self._global

    def close(self):
        self.sim.close()
