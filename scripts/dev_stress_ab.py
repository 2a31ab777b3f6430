"""Developer tool: frame time of the stress scene for several builds of the library (RMHIP_LIB), interleaved, each in a
process of its own: python scripts/dev_stress_ab.py tagA tagB ...   (raymarching_amd/variants/librmhip_<tag>.so; "main" = the
in-tree library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h, n = 1920, 1080, int(os.environ.get("RM_PRIMS", "256"))
with Material() as mat:
    Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()).bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    for k, v in scenes.STRESS_FLOATS.items(): mat.SetFloat(k, v)
    mat.SetPrimitives(scenes.stress_primitives(n))
    rt = RenderTexture(w, h, device="cuda:0")
    s = current_stream_handle("cuda:0")
    f = lambda: mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, s))
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    import hashlib
    print("%%.3f %%s" %% ((time.perf_counter() - t0) / 5 * 1e3, hashlib.sha256(rt.color.cpu().numpy().tobytes()).hexdigest()[:12]))
''' % ROOT
tags = sys.argv[1:] or ["main"]
res = {t: [] for t in tags}
for rep in range(3):
    for t in tags:
        env = dict(os.environ)
        if t != "main":
            env["RMHIP_LIB"] = os.path.join(ROOT, "raymarching_amd", "variants", f"librmhip_{t}.so")
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(t, "FAILED", out.stderr[-400:]); continue
        res[t].append(out.stdout.split())
for t in tags:
    print(t, " ".join(r[0] for r in res[t]), "ms;", set(r[1] for r in res[t]))
