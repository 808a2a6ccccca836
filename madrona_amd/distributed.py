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
        self._exec_stream = None    # torch view of the executor's stream
        self._pack_graphs = None    # step graph + pack into buffer 0 / 1
        self._slot = 0              # packed buffer of the next step
        self._exchanged = [None, None]  # per buffer: its last collective done

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
        self._packed_local = [
            torch.empty((W, offset), dtype=torch.int32, device=device)
            for _ in range(2)]
        self._packed_global = torch.empty((self.shard.total_worlds, offset),
                                          dtype=torch.int32, device=device)
        for name, dtype, tail, off, n in layout:
            view = self._packed_global[:, off:off + n].view(dtype)
            self._global[name] = view.unflatten(1, tail) if tail else view[:, 0]

    def step(self, n: int = 1):
        """Steps the local worlds, then gathers the observation tensors:
        out[name] has shape [total_worlds, ...].  On the HIP backend everything
        is queued in stream order: consume the result on the current torch
        stream, or synchronize, before reading it on the host."""
        import torch

        on_device = getattr(self.sim, "backend", "") == "hip"
        if not on_device:
            self.sim.step(n)
            if not self.obs_names:
                return self._global
            if self._local_words is None:
                self._setup()
            packed = self._packed_local[0]
            torch.cat(self._local_words, dim=1, out=packed)
            self._exchange(packed)
            return self._global

        # HIP backend: everything is stream-ordered, no host round trip.  The
        # pack is part of the replay; only the collective (which reads the
        # packed copy, not the exported columns) waits for the executor's
        # stream, and it overlaps with the next replay.  Two packed buffers
        # alternate: a buffer is packed again only after the collective that
        # read it two steps ago has finished.
        cur = torch.cuda.current_stream()
        if self._exec_stream is None:
            self._exec_stream = torch.cuda.ExternalStream(
                self.sim.stream(), device=cur.device)
        ext = self._exec_stream

        if not self.obs_names:
            self.sim.step_async(n)
            return self._global
        if self._local_words is None:
            self._setup()
        if self._pack_graphs is None:
            # the pack is the last node of the step graph itself (one graph per
            # packed buffer): a foreign kernel between two replays costs ~35 us
            # of launch pipelining, and torch.cat of eight narrow tensors ~20
            self._pack_graphs = [
                self.sim.packed_step_graph(self.obs_names, buf.data_ptr())
                for buf in self._packed_local]

        if n > 1:
            self.sim.step_async(n - 1)
        slot = self._slot
        self._slot ^= 1
        packed = self._packed_local[slot]
        if self._exchanged[slot] is not None:
            ext.wait_event(self._exchanged[slot])
        self.sim.step_async(1, graph=self._pack_graphs[slot])
        # (not cur.wait_stream(ext): an event recorded between two graph
        # launches costs ~20 us of launch pipelining; the replay's last kernel
        # bumps a counter this stream polls instead)
        self.sim.stream_wait_replays(cur.cuda_stream)
        self._exchange(packed)
        if self._exchanged[slot] is None:
            self._exchanged[slot] = torch.cuda.Event()
        self._exchanged[slot].record(cur)
        return self._global

    def _exchange(self, packed):
        if self.shard.world_size == 1:
            self._packed_global.copy_(packed)
        elif packed.is_cuda and self._dist.get_backend(self.group) == "gloo":
            # Device tensors over a host-side backend (two ranks sharing one
            # GPU in tests: RCCL refuses duplicate devices): stage through
            # pinned host memory.  Everything around the collective -- packed
            # step graphs, stream ordering, double buffering -- is the code the
            # RCCL path runs.
            import torch
            if getattr(self, "_stage", None) is None:
                self._stage = (
                    torch.empty(packed.shape, dtype=packed.dtype).pin_memory(),
                    torch.empty(self._packed_global.shape,
                                dtype=packed.dtype).pin_memory())
            send, recv = self._stage
            send.copy_(packed, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            self._dist.all_gather_into_tensor(recv, send, group=self.group)
            self._packed_global.copy_(recv, non_blocking=True)
        else:
            self._dist.all_gather_into_tensor(
                self._packed_global, packed, group=self.group)

    def sync(self):
        """Waits for queued steps (HIP backend); raises on a device error flag."""
        if getattr(self.sim, "backend", "") == "hip":
            self.sim.sync()

    def close(self):
        self.sim.close()
