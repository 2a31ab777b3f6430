"""Developer tool: the culling tolerance kernel on the stress scene against the strict tile kernel (gate numbers + time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
from helpers import assert_frame_close

w, h = int(os.environ.get("RM_W", "1920")), int(os.environ.get("RM_H", "1080"))
n_prims = int(os.environ.get("RM_N", "256"))

def run(kernel, slack=None, reps=3, fastm=0):
    with Material() as mat:
        mat.set_kernel(kernel)
        Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()).bind()
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        for k, v in scenes.STRESS_FLOATS.items(): mat.SetFloat(k, v)
        mat.SetPrimitives(scenes.stress_primitives(n_prims))
        if slack is not None: mat.SetFloat("_RmCullSlack", slack)
        mat.SetFloat("_RmCullFastMath", float(fastm))
        rt = RenderTexture(w, h, device="cuda:0")
        def step(): mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
        step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): step()
        e1.record(); torch.cuda.synchronize()
        st = mat.stats()
        return e0.elapsed_time(e1) / reps, st, {"rgba": rt.color.cpu().numpy(), "depth": rt.depth.cpu().numpy(), "hit": rt.hit.cpu().numpy()}

ms0, st0, ref = run(1)
print(f"strict tile kernel: {ms0:.2f} ms  {st0}", flush=True)
for fastm, slack in [(0, None), (0, 2.0), (0, 1.0), (0, 0.5), (1, None), (1, 1.0)]:
    ms, st, got = run(4, slack, fastm=fastm)
    n = ref["hit"].size
    mask_bad = int((got["hit"] != ref["hit"]).sum())
    same = got["hit"] == ref["hit"]
    rel = np.abs(got["depth"].astype(np.float64) - ref["depth"]) / np.maximum(np.abs(ref["depth"]), 1e-30)
    depth_bad = int((rel[same] > 1e-4).sum())
    cd = np.abs(got["rgba"].astype(np.int32) - ref["rgba"].astype(np.int32)).max(axis=-1)
    col_bad = int((cd > 1).sum())
    ok = mask_bad <= 1e-4 * n and depth_bad <= 1e-4 * n and col_bad <= 1e-3 * n
    print(f"cull fastmath {fastm} slack {slack}: {ms:.2f} ms ({ms0 / ms:.2f}x)  mask_bad {mask_bad} depth_bad {depth_bad} col_bad {col_bad} of {n}  gate {'ok' if ok else 'FAIL'}  steps_p {st['sum_steps_p']} (strict {st0['sum_steps_p']})", flush=True)
