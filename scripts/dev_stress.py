"""Developer tool: times the 256-primitive stress scene (config 5) and the rotated 4-slot scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
import cases

def timeit(mat, w, h, n=20):
    rt = RenderTexture(w, h, device="cuda:0")
    def step(): mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, mat.stats()

w, h = 1920, 1080
for kernel in (1, 2, 3):
    with Material() as mat:
        mat.set_kernel(kernel)
        rm = Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()); rm.bind()
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        for k, v in scenes.STRESS_FLOATS.items(): mat.SetFloat(k, v)
        prims = scenes.stress_primitives(256)
        mat.SetPrimitives(prims)
        ms, st = timeit(mat, w, h, n=5)
        evals = st["sum_steps_p"] + st["sum_steps_s"] + 7 * st["n_hit"]
        # algorithmic flops for N prims: scene.w = sum(47 + prim) + 13 (N-1); prim flops sphere 8 box 19 link 12 cyl 15
        per = sum(47 + [8, 19, 12, 15][p["type"]] for p in prims) + 13 * (len(prims) - 1)
        fl = (per + 11) * st["sum_steps_p"] + (per + 14) * st["sum_steps_s"] + (6 * per + 22 + 9 + (per + 36 * (len(prims) - 1)) + 6 + 9 + 7 + 4 + 11 + 9) * st["n_hit"]
        print(f"stress256 kernel {kernel}: {ms:.2f} ms/frame, hits {st['n_hit']}, evals {evals/1e6:.1f} M, {evals*256/ms/1e9:.1f} G prim-evals/s, {fl/ms/1e9:.1f} TFLOP/s algorithmic ({fl/ms/1e9/157.3*100:.1f} %)")
    with Material() as mat:
        mat.set_kernel(kernel)
        rm = Raymarching(scenes.default_camera(w, h), mat, cases.transformed_transforms(False)); rm.bind()
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        ms, st = timeit(mat, w, h, n=50)
        fl = 292.0 * st["sum_steps_p"] + 295.0 * st["sum_steps_s"] + 2152.0 * st["n_hit"]
        print(f"rotated+scaled 4-slot scene kernel {kernel}: {ms*1000:.1f} us/frame, {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.1f} %)")
