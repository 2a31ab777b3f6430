"""Multi-GPU sharding of the march: one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm), frames sharded by camera pose or by interleaved row bands,
finished pixels gathered to a root rank over xGMI.

Pixels and poses are independent, so rendering needs no communication; the only
exchange step is the gather of finished frames to the rank that displays or
encodes them.  Row bands are interleaved (band b -> rank b mod N) rather than
contiguous because the sample scene's geometry sits in one half of the image.

The renderer is a callback so that the sharding/gather logic runs unchanged with
the gloo backend on CPU tensors in the tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def band_rows_of(h: int, band_rows: int, part: int, n_parts: int) -> List[int]:
    """frame rows (increasing) owned by `part` under interleaved band sharding"""
    rows: List[int] = []
    n_bands = (h + band_rows - 1) // band_rows
    for b in range(part, n_bands, n_parts):
        rows.extend(range(b * band_rows, min(h, (b + 1) * band_rows)))
    return rows


def poses_of(n_poses: int, rank: int, world: int) -> List[int]:
    """pose k -> rank k mod world"""
    return list(range(rank, n_poses, world))


def _gather_to_root(t: torch.Tensor, root: int, group=None) -> Optional[List[torch.Tensor]]:
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [t]
    out = [torch.empty_like(t) for _ in range(world)] if rank == root else None
    dist.gather(t, out, dst=root, group=group)
    return out


def render_frame_bands(render_bands: Callable[[int, int, int], Sequence[torch.Tensor]], h: int, band_rows: int,
                       root: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """One frame, row bands interleaved over the ranks.

    render_bands(band_rows, part, n_parts) returns this rank's compact planes (each
    [rows, ...]).  Returns the reassembled full-frame planes on the root, None elsewhere.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    planes = list(render_bands(band_rows, rank, world))
    max_rows = max(len(band_rows_of(h, band_rows, p, world)) for p in range(world))
    full = []
    for pl in planes:
        pad = torch.zeros((max_rows,) + tuple(pl.shape[1:]), dtype=pl.dtype, device=pl.device)
        pad[: pl.shape[0]] = pl
        got = _gather_to_root(pad, root, group)
        if rank == root:
            out = torch.empty((h,) + tuple(pl.shape[1:]), dtype=pl.dtype, device=pl.device)
            for p in range(world):
                rows = band_rows_of(h, band_rows, p, world)
                if rows:
                    out[torch.as_tensor(rows, device=pl.device)] = got[p][: len(rows)]
            full.append(out)
    return full if rank == root else None


def render_poses(render_pose_batch: Callable[[List[int]], torch.Tensor], n_poses: int, root: int = 0,
                 group=None) -> Optional[torch.Tensor]:
    """n_poses frames, pose k on rank k mod world.  render_pose_batch(pose_ids) returns
    [len(pose_ids), h, w, c].  Returns [n_poses, h, w, c] in pose order on the root."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = poses_of(n_poses, rank, world)
    per = (n_poses + world - 1) // world
    local = render_pose_batch(mine)
    if local.shape[0] < per:
        pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        local = pad
    got = _gather_to_root(local.contiguous(), root, group)
    if rank != root:
        return None
    out = torch.empty((n_poses,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for p in range(world):
        ids = poses_of(n_poses, p, world)
        if ids:
            out[torch.as_tensor(ids, device=local.device)] = got[p][: len(ids)]
    return out


class PipelinedGather:
    """Gathers finished frames to the root on a side stream while later steps render, with
    `depth` buffers of `batch` frames each in rotation.  One collective moves `batch` frames:
    a frame of the sample scene renders in under 100 us, less than the host needs to issue a
    collective, so gathering frame by frame would make the host the bottleneck.

        g = PipelinedGather(frame_like, depth=2, batch=8)
        for step in range(K):
            buf = g.acquire()          # slot for this step's frame; waits (stream-side) for the
                                       # buffer's previous gather when a new buffer is started
            render_into(buf)           # on the current stream
            g.submit(buf)              # every `batch`-th call gathers the buffer on the side stream
        g.drain()                      # gathers a partly filled buffer too, then joins the side stream
    """

    def __init__(self, like: torch.Tensor, root: int = 0, group=None, depth: int = 2, single_rank_too: bool = False,
                 batch: int = 1):
        assert depth >= 1 and batch >= 1
        self.root, self.group = root, group
        self.single_rank_too = single_rank_too      # run the collective even with one rank (exercises the stream logic)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.batch = batch
        shape = (batch,) + tuple(like.shape)
        self.bufs = [torch.empty(shape, dtype=like.dtype, device=like.device) for _ in range(depth)]
        self.cuda = like.is_cuda
        self.side = torch.cuda.Stream(device=like.device) if self.cuda else None
        self.done = [None] * depth          # event: gather of bufs[i] finished
        self.recv = None
        if self.rank == root and (self.world > 1 or single_rank_too):
            self.recv = [[torch.empty(shape, dtype=like.dtype, device=like.device) for _ in range(self.world)]
                         for _ in range(depth)]
        self.i = 0                          # frames submitted so far
        self.last = None                    # (buffer index, frames in it) of the most recent gather

    def acquire(self) -> torch.Tensor:
        k, j = (self.i // self.batch) % len(self.bufs), self.i % self.batch
        if j == 0 and self.cuda and self.done[k] is not None:
            torch.cuda.current_stream().wait_event(self.done[k])
        return self.bufs[k][j]

    def _gather(self, k: int, n_frames: int) -> None:
        self.last = (k, n_frames)
        if self.world == 1 and not self.single_rank_too:
            return
        buf = self.bufs[k]
        if not self.cuda:
            dist.gather(buf, self.recv[k] if self.recv else None, dst=self.root, group=self.group)
            return
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            dist.gather(buf, self.recv[k] if self.recv else None, dst=self.root, group=self.group)
            ev = torch.cuda.Event()
            ev.record()
            self.done[k] = ev

    def submit(self, buf: torch.Tensor) -> None:
        k = (self.i // self.batch) % len(self.bufs)
        self.i += 1
        if self.i % self.batch == 0:
            self._gather(k, self.batch)

    def last_gathered(self) -> Optional[List[torch.Tensor]]:
        """root only: per rank, the frames [n, ...] of the most recent gather (valid after drain())"""
        if self.recv is None or self.last is None:
            return None
        k, n = self.last
        return [t[:n] for t in self.recv[k]]

    def drain(self) -> None:
        part = self.i % self.batch
        if part:
            # a partly filled buffer travels whole (its unused slots hold stale frames)
            self._gather((self.i // self.batch) % len(self.bufs), part)
            self.i += self.batch - part
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.side)
