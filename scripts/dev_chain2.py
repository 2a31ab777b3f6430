"""Developer probe: the longest 8x8 tiles of the 1080p sample frame (tile coordinates from the CPU simulation,
scripts/dev_chain_skip_sim.py), each rendered alone as an 8x8 frame whose frustum corners are the tile's, with the exact pair skip
of the shadow march on and off; then the whole frame both ways."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle
from raymarching_amd.unity import Matrix4x4
W, H = 1920, 1080
dev = "cuda:0"

def timed(mat, w, h, n=200):
    depth = torch.empty((h, w), dtype=torch.float32, device=dev); hit = torch.empty((h, w), dtype=torch.uint8, device=dev)
    frame = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
    f = lambda: mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(frame), _ptr(depth), _ptr(hit), w, h, current_stream_handle(dev)))
    for _ in range(20): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, mat.stats(), frame.cpu().numpy()

mat = Material(device=0)
Raymarching(scenes.default_camera(W, H), mat, scenes.default_transforms()).bind()
mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
full_fc = mat.GetMatrix("_FrustrumCorners")
corners = full_fc.m.copy().reshape(4, 4).T.copy()
TL, TR, BR, BL = corners
def crop(tx, ty):
    at = lambda u, v: BL + u * (BR - BL) + v * (TL - BL)
    u0, u1, v0, v1 = tx * 8 / W, (tx * 8 + 8) / W, ty * 8 / H, (ty * 8 + 8) / H
    m = np.stack([at(u0, v1), at(u1, v1), at(u1, v0), at(u0, v0)]).astype(np.float32)
    M = Matrix4x4()
    for i in range(4): M.SetRow(i, m[i])
    mat.SetMatrix("_FrustrumCorners", M)
for tx, ty in ((86, 55), (103, 62), (88, 55), (88, 36)):
    crop(tx, ty)
    res = []
    for skip in (1.0, 0.0):
        mat.SetFloat("_RmPairSkip", skip)
        t, s, fr = timed(mat, 8, 8, 100)
        res.append((t, s["sum_steps_p"], s["sum_steps_s"], fr))
    print("tile (%d,%d) alone: pair skip on %.1f us, off %.1f us; steps p %d s %d; same frame %s" %
          (tx, ty, res[0][0], res[1][0], res[0][1], res[0][2], np.array_equal(res[0][3], res[1][3])))
mat.SetMatrix("_FrustrumCorners", full_fc)
for skip in (1.0, 0.0, 1.0, 0.0):
    mat.SetFloat("_RmPairSkip", skip)
    t, s, _ = timed(mat, W, H, 300)
    print("whole frame, pair skip %d: %.1f us" % (skip, t))
