"""Developer tool: where sharing a tile between four waves starts to pay: 1080p frames of small tables on the one-wave tile
kernel (1) and on the shared-tile kernel (5), with a build whose RM_COOP_MIN_PRIMS is 3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = 1920, 1080
for n in (3, 4, 5, 6, 7, 8, 9, 12, 13, 16, 20, 24, 28, 32, 48):
    res = {}
    for kernel in (1, 5):
        with Material() as mat:
            mat.set_kernel(kernel)
            Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()).bind()
            mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            for k, v in scenes.STRESS_FLOATS.items():
                mat.SetFloat(k, v)
            prims = scenes.stress_primitives(n, seed=0x5EED0000 + n)
            if os.environ.get("RM_UNTRANSFORMED"):
                prims = [dict(p, quat=(0.0, 0.0, 0.0, 1.0), scale=(1.0, 1.0, 1.0)) for p in prims]
            mat.SetPrimitives(prims)
            rt = RenderTexture(w, h, device="cuda:0")
            step = lambda: mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
            step(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                step()
            e1.record(); torch.cuda.synchronize()
            res[kernel] = (e0.elapsed_time(e1) / 10, rt.color.cpu().numpy().copy())
    same = np.array_equal(res[1][1], res[5][1])
    print(f"n = {n}: one wave per tile {res[1][0]:.3f} ms, four waves per tile {res[5][0]:.3f} ms ({res[1][0] / res[5][0]:.2f}x), identical {same}", flush=True)
