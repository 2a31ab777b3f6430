"""The multi-GPU sharding layer with two gloo ranks on CPU: row-band and pose sharding
plus the gather to the root must reproduce the one-process frame byte for byte.  The
renderer plugged in here is the CPU oracle (this is a test of the sharding/gather
logic; the HIP renderer is exercised by the GPU tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rm_oracle as O
        from raymarching_amd import Material, Raymarching, scenes
        from raymarching_amd import distributed as D

        w, h, band = 160, 90, 8
        with Material() as mat:
            rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms())
            rm.bind()
            mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            props = O.props_from_material(mat)
        uni = O.Uniforms(props)
        full = O.render(uni, w, h, threads=2)

        # ---- one frame, interleaved row bands -------------------------------------------
        def render_bands(band_rows, part, n_parts):
            rows = D.band_rows_of(h, band_rows, part, n_parts)
            rgba = np.zeros((len(rows), w, 4), dtype=np.uint8)
            depth = np.zeros((len(rows), w), dtype=np.float32)
            # the oracle renders row ranges of the full frame; copy this rank's rows out compactly
            i = 0
            while i < len(rows):
                j = i
                while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1:
                    j += 1
                o = O.render(uni, w, h, y0=rows[i], y1=rows[j] + 1, threads=1)
                rgba[i:j + 1] = o["rgba"][rows[i]:rows[j] + 1]
                depth[i:j + 1] = o["depth"][rows[i]:rows[j] + 1]
                i = j + 1
            return [torch.from_numpy(rgba), torch.from_numpy(depth)]

        got = D.render_frame_bands(render_bands, h, band, root=0)
        if rank == 0:
            assert np.array_equal(got[0].numpy(), full["rgba"]), "band-sharded colour differs"
            assert np.array_equal(got[1].numpy().view(np.uint32), full["depth"].view(np.uint32)), "band-sharded depth differs"
        else:
            assert got is None

        # ---- 5 poses over 2 ranks ---------------------------------------------------------
        n_poses = 5

        def pose_props(k):
            with Material() as m:
                r = Raymarching(scenes.orbit_camera(k * 7, 64, w, h), m, scenes.default_transforms())
                r.bind()
                m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
                return O.props_from_material(m)

        def render_pose_batch(ids):
            out = np.zeros((len(ids), h, w, 4), dtype=np.uint8)
            for i, k in enumerate(ids):
                out[i] = O.render(O.Uniforms(pose_props(k)), w, h, threads=1)["rgba"]
            return torch.from_numpy(out)

        frames = D.render_poses(render_pose_batch, n_poses, root=0)
        if rank == 0:
            assert frames.shape == (n_poses, h, w, 4)
            for k in range(n_poses):
                ref = O.render(O.Uniforms(pose_props(k)), w, h, threads=2)["rgba"]
                assert np.array_equal(frames[k].numpy(), ref), f"pose {k} differs after gather"

        # ---- pipelined gather (the bench's overlap helper) on CPU tensors ----------------
        like = torch.zeros((4, 3), dtype=torch.uint8)
        pipe = D.PipelinedGather(like, root=0, depth=2)
        for step in range(3):
            buf = pipe.acquire()
            buf.fill_(10 * step + rank)
            pipe.submit(buf)
        pipe.drain()
        if rank == 0:
            last = pipe.last_gathered()
            assert [int(t[0, 0, 0]) for t in last] == [20, 21]
        # three frames per collective; the fourth and fifth frame leave with drain() as a partial batch
        pipe = D.PipelinedGather(like, root=0, depth=2, batch=3)
        for step in range(5):
            buf = pipe.acquire()
            buf.fill_(10 * step + rank)
            pipe.submit(buf)
            if step == 2 and rank == 0:
                assert [[int(f[0, 0]) for f in t] for t in pipe.last_gathered()] == [[0, 10, 20], [1, 11, 21]]
        pipe.drain()
        if rank == 0:
            last = pipe.last_gathered()
            assert [[int(f[0, 0]) for f in t] for t in last] == [[30, 40], [31, 41]]
        # ---- compact tile transport: hit tiles only, variable-size point-to-point gather -------------
        from helpers import RECORD_WORDS, compact_frames_numpy, expand_frames_numpy
        n_poses = 6
        mine = D.poses_of(n_poses, rank, world)
        rendered = [O.render(O.Uniforms(pose_props(k)), w, h, threads=1) for k in mine]
        frames = torch.from_numpy(np.stack([r["rgba"] for r in rendered]))
        hits = torch.from_numpy(np.stack([r["hit"] for r in rendered]))

        def compact(fr, ht, records, count):
            rec = compact_frames_numpy(fr.numpy(), ht.numpy())
            records[: rec.shape[0]] = torch.from_numpy(rec)
            count[0] = rec.shape[0]

        tiles = ((w + 7) // 8) * ((h + 7) // 8)
        tg = D.TileGather(compact, capacity_records=len(mine) * tiles, device="cpu", root=0)
        bg = np.array([49, 77, 121, 255], dtype=np.uint8)          # unorm8 of the camera's background colour
        delivered = {}

        def on_received(p, records, n):
            ids = D.poses_of(n_poses, p, world)
            rec = records[:n].numpy() if n else np.zeros((0, RECORD_WORDS), dtype=np.int32)
            out = expand_frames_numpy(rec, len(ids), h, w, bg)
            for i, k in enumerate(ids):
                delivered[k] = out[i]

        for rep in range(2):                                        # twice: the record buffers rotate
            delivered.clear()
            t = tg.submit(frames, hits)
            counts = tg.complete(t, on_received)
            tg.drain()
            assert len(counts) == world and 0 < counts[rank] < len(mine) * tiles
            if rank == 0:
                assert all(0 < c < len(mine) * tiles for c in counts)         # the root is told every count, a peer only knows its own
                assert sorted(delivered) == list(range(n_poses))
                for k in range(n_poses):
                    ref = O.render(O.Uniforms(pose_props(k)), w, h, threads=2)["rgba"]
                    assert np.array_equal(delivered[k], ref), f"pose {k} differs after the compact gather"
            else:
                assert all(c == -1 for i, c in enumerate(counts) if i != rank)
                assert not delivered and tg.bytes_sent == (rep + 1) * counts[rank] * RECORD_WORDS * 4
        # ---- the root keeps its dense frames: later deliveries into a slot update the tiles that changed ----------
        from helpers import update_frames_numpy
        dense = {p: np.zeros((len(D.poses_of(n_poses, p, world)), h, w, 4), dtype=np.uint8) for p in range(world)}
        calls = []

        def expand_np(rec, n, dst):
            calls.append("expand")
            dst[...] = expand_frames_numpy(rec[:n].numpy(), dst.shape[0], h, w, bg)

        def update_np(old, n_old, rec, n, dst):
            calls.append("update")
            update_frames_numpy(dst, old[:n_old].numpy(), rec[:n].numpy(), bg)

        store = D.DenseFrames(expand_np, update_np)
        for rep, shift in enumerate((0, 7, 3)):                       # three deliveries of different poses into the same slots
            rendered = [O.render(O.Uniforms(pose_props(k + shift)), w, h, threads=1) for k in mine]
            fr = torch.from_numpy(np.stack([r["rgba"] for r in rendered]))
            ht = torch.from_numpy(np.stack([r["hit"] for r in rendered]))
            t = tg.submit(fr, ht)
            tg.complete(t, lambda p, records, n: store.deliver(p, records.clone(), n, dense[p]))
            tg.drain()
            if rank == 0:
                assert calls == ["expand"] * world + ["update"] * (world * rep)
                for p in range(world):
                    for i, k in enumerate(D.poses_of(n_poses, p, world)):
                        ref = O.render(O.Uniforms(pose_props(k + shift)), w, h, threads=2)["rgba"]
                        assert np.array_equal(dense[p][i], ref), f"delivery {rep}: pose {k + shift} differs after the update"
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_band_and_pose_sharding_two_ranks(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_band_rows_match_the_c_abi():
    """the Python sharding helper and rm_band_rows() of the library agree"""
    sys.path.insert(0, ROOT)
    from raymarching_amd import distributed as D
    from raymarching_amd import native
    lib = native.lib()
    for h, b, n in [(1080, 8, 8), (2160, 8, 2), (37, 5, 3), (7, 8, 2), (4320, 16, 8)]:
        rows_all = []
        for p in range(n):
            rows = D.band_rows_of(h, b, p, n)
            assert len(rows) == lib.rm_band_rows(h, b, p, n)
            rows_all += rows
        assert sorted(rows_all) == list(range(h))
    assert D.poses_of(64, 3, 8) == list(range(3, 64, 8))
