"""Developer probe: the 3840x2160 directional depth pass (ShadowMapDirectional's rectangle) followed by the
frag pass, timed separately and together on one stream."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases                                                                        # noqa: E402
from raymarching_amd import Material                                                # noqa: E402
from raymarching_amd.material import _fp, _ptr, current_stream_handle               # noqa: E402

W, H = 3840, 2160
dev = "cuda:0"
case = cases.BY_NAME["shadow_dir_3840x2160"]


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with Material() as mat:
    case.bind(mat)
    c, f = np.ascontiguousarray(case.corners), np.ascontiguousarray(case.forward)
    smap = torch.empty((H, W), dtype=torch.float32, device=dev)
    shit = torch.empty((H, W), dtype=torch.uint8, device=dev)
    color = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    depth = torch.empty((H, W), dtype=torch.float32, device=dev)
    hit = torch.empty((H, W), dtype=torch.uint8, device=dev)
    lib, h, st = mat._lib, mat.handle, current_stream_handle(dev)

    def shadow():
        mat._ck(lib.rm_shadowmap_directional_device(h, _fp(c), _fp(f), _ptr(smap), _ptr(shit), W, H, st))

    def frag():
        mat._ck(lib.rm_blit_device(h, None, None, _ptr(color), _ptr(depth), _ptr(hit), W, H, st))

    def both():
        shadow()
        frag()
    for kern, name in ((0, "auto"), (1, "tile"), (2, "queue")):
        mat.set_kernel(kern)
        ts = timed(shadow)
        s_stats = mat.stats()
        tf = timed(frag)
        f_stats = mat.stats()
        tb = timed(both)
        rays = s_stats["n_pixels"] + f_stats["n_pixels"] + f_stats["n_hit"]
        print("%s: depth pass %.1f us (%d evals), frag %.1f us, both %.1f us = %.0f Mrays/s"
              % (name, ts, s_stats["sum_steps_p"], tf, tb, rays / tb))
