#!/usr/bin/env python3
"""Developer tool: what the gather costs rank 0 of an 8-GPU run of the batched-pose configuration, on one GPU.  Per step
the main stream renders this rank's 8 poses (two batches of 4 at 7680x4320); a side stream expands the records of the 56
poses the other ranks would have sent (stand-ins: this rank's own records, reused) into dense frames, either whole
(rm_expand_frames_device) or by updating the tiles that changed (rm_update_frames_device).
    python scripts/dev_root_overlap.py [blocks_per_cu ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from raymarching_amd import Material, Raymarching, scenes  # noqa: E402
from raymarching_amd.distributed import RECORD_WORDS  # noqa: E402
from raymarching_amd.material import current_stream_handle  # noqa: E402

dev = "cuda:0"
w, h = bench.POSE_W, bench.POSE_H
tiles = ((w + 7) // 8) * ((h + 7) // 8)


def run(cap, expand, steps=4):
    mat = Material(device=0)
    Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()).bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    if cap:
        mat.SetFloat("_RmBlocksPerCU", float(cap))
    frames = [bench.Frame(torch, dev, w, h, 4) for _ in range(2)]
    arrays = [bench.pose_arrays(scenes, Raymarching, mat, list(range(b * 4, b * 4 + 4)), w, h) for b in range(2)]
    records = torch.empty((4 * tiles, RECORD_WORDS), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    dense = torch.empty((4, h, w, 4), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    stream = current_stream_handle(dev)
    fcs, c2ws, pos = arrays[0]
    bench.blit_poses(mat, fcs, c2ws, pos, frames[0], w, h, stream)
    mat.compact_frames(frames[0].color, frames[0].hit, records, count)
    n = int(count.item())
    peer = records[:n].clone()                               # what a peer sent: not touched by this rank's compactions

    def step():
        for b in range(2):
            fcs, c2ws, pos = arrays[b]
            bench.blit_poses(mat, fcs, c2ws, pos, frames[b], w, h, stream)
            mat.compact_frames(frames[b].color, frames[b].hit, records, count)
            if expand:
                with torch.cuda.stream(side):
                    for _ in range(7):                       # 7 peers x 4 poses per batch
                        if expand == 1:
                            mat.expand_frames(peer, n, dense, stream=side.cuda_stream)
                        else:
                            mat.update_frames(peer, n, peer, n, dense, stream=side.cuda_stream)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mat.close()
    return dt, n


for cap in [int(a) for a in sys.argv[1:]] or [0, 3]:
    a, n = run(cap, 0)
    b, _ = run(cap, 1)
    c, _ = run(cap, 2)
    print(f"blocks/CU cap {cap}: 8 poses {a * 1e3:.2f} ms; with 56 full expansions ({n} records per 4 poses) on a side stream {b * 1e3:.2f} ms; "
          f"with 56 updates (old tiles restored, new records written) {c * 1e3:.2f} ms", flush=True)
