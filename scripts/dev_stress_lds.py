"""Developer tool: stress scene (256 primitives) with the table in LDS vs read through the scalar cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = 1920, 1080
for n in (256, 64, 16):
    for lds in (1.0, 0.0):
        with Material() as mat:
            mat.set_kernel(1)
            rm = Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()); rm.bind()
            mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            for k, v in scenes.STRESS_FLOATS.items(): mat.SetFloat(k, v)
            mat.SetFloat("_RmLdsTable", lds)
            mat.SetPrimitives(scenes.stress_primitives(n))
            rt = RenderTexture(w, h, device="cuda:0")
            def step(): mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
            for _ in range(2): step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): step()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            st = mat.stats()
            evals = st["sum_steps_p"] + st["sum_steps_s"] + 7 * st["n_hit"]
            print(f"n {n} lds {lds}: {ms:.2f} ms/frame, {evals*n/ms/1e6:.1f} G prim-evals/s")
