"""Developer tool: times one named parity case at a chosen resolution with each kernel.
    python scripts/dev_case.py menger_closeup_320x180 1920 1080 [flops_per_eval]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from raymarching_amd import Material, RenderTexture
from raymarching_amd.material import _ptr, current_stream_handle
import cases

name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fpe = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
c = cases.BY_NAME[name]
c.w, c.h = w, h
for kernel in (1, 2):
    with Material() as mat:
        c.bind(mat)
        mat.set_kernel(kernel)
        rt = RenderTexture(w, h, device="cuda:0")
        def step(): mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
        for _ in range(5): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n): step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        st = mat.stats()
        evals = st["sum_steps_p"] + st["sum_steps_s"] + 7 * st["n_hit"]
        rays = st["n_pixels"] + st["n_hit"]
        print(f"{name} {w}x{h} kernel {kernel}: {ms*1e3:.1f} us/frame, {rays/ms/1e3:.0f} Mrays/s, {evals/ms/1e6:.2f} G evals/s, hits {st['n_hit']}, evals {evals}"
              + (f", {evals*fpe/ms/1e9:.1f} TFLOP/s literal ({evals*fpe/ms/1e9/157.3*100:.1f} %)" if fpe else ""))
