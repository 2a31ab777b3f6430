"""Developer probe for the batched-pose configuration: one GPU's share of 64 orbit poses at 7680x4320
(pose k on rank k mod 8 -> 8 poses) in ONE rm_blit_poses_device call.  Checks pose 17 against the
committed hashes and prints the throughput next to eight single blits."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raymarching_amd import Material, Raymarching, RenderTexture, scenes          # noqa: E402
from raymarching_amd.material import _fp, _ptr, current_stream_handle               # noqa: E402

W, H = 7680, 4320
KS = [1, 9, 17, 25, 33, 41, 49, 57]          # rank 1 of 8
dev = "cuda:0"
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["cases"]["orbit_k17_7680x4320"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


with Material() as mat:
    fcs, c2ws, pos = [], [], []
    rt = RenderTexture(W, H, device=dev)
    t_single = 0.0
    for k in KS:
        rm = Raymarching(scenes.orbit_camera(k, 64, W, H), mat, scenes.default_transforms())
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        rm.OnRenderImage(None, rt)                      # warm
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            rm.OnRenderImage(None, rt)
        e1.record()
        torch.cuda.synchronize()
        t_single += e0.elapsed_time(e1) / 5
        fcs.append(rm.GetFrustumCorners().m)
        c2ws.append(rm.cam.cameraToWorldMatrix.m)
        pos.append(rm.cam.transform.position)
    n = len(KS)
    fcs = np.ascontiguousarray(np.stack(fcs), dtype=np.float32)
    c2ws = np.ascontiguousarray(np.stack(c2ws), dtype=np.float32)
    pos = np.ascontiguousarray(np.stack(pos), dtype=np.float32)
    color = torch.empty((n, H, W, 4), dtype=torch.uint8, device=dev)
    depth = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    hit = torch.empty((n, H, W), dtype=torch.uint8, device=dev)

    def batch():
        mat._ck(mat._lib.rm_blit_poses_device(mat.handle, n, _fp(fcs), _fp(c2ws), _fp(pos), _ptr(color), _ptr(depth), _ptr(hit),
                                              W, H, current_stream_handle(dev)))
    for _ in range(2):
        batch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        batch()
    e1.record()
    torch.cuda.synchronize()
    t_batch = e0.elapsed_time(e1) / 5
    st = mat.stats()
    rays = st["n_pixels"] + st["n_hit"]
    i = KS.index(17)
    ok = (sha(color[i].cpu().numpy()) == gold["sha_rgba"] and sha(depth[i].cpu().numpy()) == gold["sha_depth"]
          and sha(hit[i].cpu().numpy()) == gold["sha_hit"])
    print("8 poses at %dx%d: one call %.2f ms (%.0f Mrays/s), eight single blits %.2f ms; pose 17 matches the committed hashes: %s"
          % (W, H, t_batch, rays / t_batch / 1e3, t_single, ok))
    print(st)
