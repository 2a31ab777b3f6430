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


RECORD_WORDS = 49          # one compact-transport record: tile index + 64 pixels as RGB8 (rm_tile_record_bytes() / 4)


class _Ticket:
    __slots__ = ("slot", "n_frames", "count_dev", "count_host", "count_ready", "tag")


class TileGather:
    """Gathers finished frames to the root as compact tile records (rm_compact_frames_device): only the 8x8
    tiles that hold a hit pixel travel, everything else the root reconstructs from the background it already
    has (rm_expand_frames_device).  Record counts differ per rank and per batch, so every peer first tells the root its
    count (one small host message on `count_group`, gloo; the peers do not wait for each other or for the root -- if that
    group is RCCL too, the counts are all-gathered as device tensors instead), then exactly count records travel
    point-to-point from every peer to the root (`group`: RCCL over xGMI on GPUs, each peer on its own link).

        g = TileGather(compact, capacity_records, like_device, root=0)
        t = g.submit(frames, hit)              # compaction on the current stream; returns at once
        ... enqueue the next batch's rendering ...
        g.complete(t, on_received)             # waits (host) for this batch's count, then transfers on a side
                                               # stream; on the root on_received(rank, records, n) runs there
        g.drain()

    compact(frames, hit, records_out, count_out) must enqueue the compaction on the current stream.
    """

    def __init__(self, compact: Callable, capacity_records: int, device, root: int = 0, group=None, count_group=None,
                 depth: int = 2):
        self.compact, self.root, self.group = compact, root, group
        self.count_group = count_group if count_group is not None else group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.records = [torch.empty((capacity_records, RECORD_WORDS), dtype=torch.int32, device=self.device) for _ in range(depth)]
        self.count_dev = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(depth)]
        self.count_host = [torch.zeros(1, dtype=torch.int32).pin_memory() if self.cuda else torch.zeros(1, dtype=torch.int32)
                           for _ in range(depth)]
        self.free_ev = [None] * depth          # event: the transfer out of records[i] has finished
        self.i = 0
        self.bytes_sent = 0                    # payload bytes this rank has sent (diagnostics)
        self.last_counts: Optional[List[int]] = None
        self._count_send = None                # (work handle, tensor) of this peer's count message still in flight

    def submit(self, frames: torch.Tensor, hit: torch.Tensor, tag=None) -> _Ticket:
        k = self.i % len(self.records)
        self.i += 1
        if self.cuda and self.free_ev[k] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.free_ev[k])
        self.compact(frames, hit, self.records[k], self.count_dev[k])
        t = _Ticket()
        t.slot, t.n_frames, t.tag = k, int(frames.shape[0]), tag
        t.count_dev, t.count_host = self.count_dev[k], self.count_host[k]
        if self.cuda:
            t.count_host.copy_(t.count_dev, non_blocking=True)
            t.count_ready = torch.cuda.Event()
            t.count_ready.record()
        else:
            t.count_host.copy_(t.count_dev)
            t.count_ready = None
        return t

    def complete(self, t: _Ticket, on_received: Optional[Callable[[int, torch.Tensor, int], None]] = None) -> List[int]:
        """returns the record counts of this batch, indexed by rank: all of them on the root; on a peer its own, -1 for the
        others (only the root is told)"""
        if t.count_ready is not None:
            t.count_ready.synchronize()
        mine = int(t.count_host.item())
        if self.world > 1 and self.cuda and dist.get_backend(self.count_group) == "nccl":
            # no host-side group: the counts travel as device tensors, on the side stream so that they do not queue
            # behind rendering that is already enqueued on the current one
            with torch.cuda.stream(self.side):
                outs = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(self.world)]
                dist.all_gather(outs, torch.tensor([mine], dtype=torch.int32, device=self.device), group=self.count_group)
                counts = [int(c) for c in torch.cat(outs).tolist()]
        elif self.world > 1:
            # Only the root needs the counts: every peer posts its own to the root and goes on (no rendezvous among the
            # peers, no wait for the slowest rank's batch); the root collects them, in rank order.
            counts = [-1] * self.world
            counts[self.rank] = mine
            if self.rank == self.root:
                for p in range(self.world):
                    if p != self.root:
                        c = torch.zeros(1, dtype=torch.int32)
                        dist.recv(c, self._global(p, self.count_group), group=self.count_group)
                        counts[p] = int(c.item())
            else:
                if self._count_send is not None:
                    self._count_send[0].wait()                      # the previous batch's (long gone; frees its tensor)
                msg = torch.tensor([mine], dtype=torch.int32)
                self._count_send = (dist.isend(msg, self._global(self.root, self.count_group), group=self.count_group), msg)
        else:
            counts = [mine]
        self.last_counts = counts
        rec = self.records[t.slot]

        def transfer_through_host():
            # no device-to-device transport in this group (gloo with device tensors: a rehearsal on one GPU, or a node
            # without RCCL peer access): the records cross as host tensors, blocking, peers in rank order
            bufs = {}
            if self.rank == self.root:
                for p in range(self.world):
                    if p != self.root and counts[p] > 0:
                        h = torch.empty((counts[p], RECORD_WORDS), dtype=torch.int32)
                        dist.recv(h, self._global(p), group=self.group)
                        bufs[p] = h.to(self.device)
            elif mine > 0:
                dist.send(rec[:mine].cpu(), self._global(self.root), group=self.group)
                self.bytes_sent += mine * RECORD_WORDS * 4
            if self.rank == self.root and on_received is not None:
                for p in range(self.world):
                    on_received(p, rec if p == self.root else bufs.get(p), mine if p == self.root else counts[p])

        def transfer():
            if self.cuda and self.world > 1 and dist.get_backend(self.group) == "gloo":
                return transfer_through_host()
            ops, bufs = [], {}
            if self.rank == self.root:
                for p in range(self.world):
                    if p != self.root and counts[p] > 0:
                        bufs[p] = torch.empty((counts[p], RECORD_WORDS), dtype=torch.int32, device=self.device)
                        ops.append(dist.P2POp(dist.irecv, bufs[p], self._global(p), group=self.group))
            elif mine > 0:
                ops.append(dist.P2POp(dist.isend, rec[:mine], self._global(self.root), group=self.group))
                self.bytes_sent += mine * RECORD_WORDS * 4
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if self.rank == self.root and on_received is not None:
                for p in range(self.world):
                    if p == self.root:
                        on_received(p, rec, mine)
                    else:
                        on_received(p, bufs.get(p), counts[p])

        if self.cuda:
            with torch.cuda.stream(self.side):
                # the batch's compaction and nothing later: rendering enqueued since submit() must not hold the transfer up
                self.side.wait_event(t.count_ready)
                transfer()
                ev = torch.cuda.Event()
                ev.record()
                self.free_ev[t.slot] = ev
        else:
            transfer()
        return counts

    def _global(self, group_rank: int, group="data") -> int:
        g = self.group if group == "data" else group
        return dist.get_global_rank(g, group_rank) if g is not None else group_rank

    def drain(self) -> None:
        if self._count_send is not None:
            self._count_send[0].wait()
            self._count_send = None
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.side)


class DenseFrames:
    """The displaying rank's dense frames, kept from one delivery to the next.  The first delivery into a slot builds the
    frames whole (expand: rm_expand_frames_device); later ones only put the background back where the slot's previous
    records wrote and write the new records (update: rm_update_frames_device), so the root's memory traffic per step is
    the hit tiles', not the frames'.

        store = DenseFrames(expand=lambda rec, n, dst: mat.expand_frames(rec, n, dst),
                            update=lambda old, n_old, rec, n, dst: mat.update_frames(old, n_old, rec, n, dst))
        store.deliver(key, records, n, dst)       # e.g. from TileGather's on_received; key names the slot (rank, batch)

    The store keeps a reference to the records of the latest delivery per slot; callers must not overwrite them.
    """

    def __init__(self, expand: Callable, update: Optional[Callable] = None):
        self.expand, self.update = expand, update
        self.shown = {}

    def deliver(self, key, records, n: int, dst) -> None:
        prev = self.shown.get(key)
        if prev is not None and self.update is not None:
            self.update(prev[0], prev[1], records, n, dst)
        else:
            self.expand(records, n, dst)
        self.shown[key] = (records, n)

    def forget(self, key=None) -> None:
        """the frames of `key` (or all) were overwritten by something else: the next delivery builds them whole"""
        if key is None:
            self.shown.clear()
        else:
            self.shown.pop(key, None)
