"""Developer tool: the 256-primitive stress scene at 1080p -- frame time, counters, what the exact culling executed, and the
step-count planes' distribution per 8x8 tile (how uneven the tiles are)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = int(os.environ.get("RM_W", "1920")), int(os.environ.get("RM_H", "1080"))
n = int(os.environ.get("RM_PRIMS", "256"))
with Material() as mat:
    mat.set_kernel(int(os.environ.get("RM_KERNEL", "0")))
    rm = Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()); rm.bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    for k, v in scenes.STRESS_FLOATS.items(): mat.SetFloat(k, v)
    mat.SetPrimitives(scenes.stress_primitives(n))
    rt = RenderTexture(w, h, device="cuda:0")
    s = current_stream_handle("cuda:0")
    def frame():
        mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, s))
    frame(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): frame()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    st = mat.stats()
    print("ms per frame %.2f" % ms, st)
    try:
        ls = mat.leaf_stats()
        print("leaf evaluations executed %.3e of %.3e = %.3f" % (ls["visited"], ls["possible"], ls["visited"] / ls["possible"]))
    except Exception as e:
        print("no leaf stats:", e)
    steps = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    mat._ck(mat._lib.rm_blit_steps_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), _ptr(steps), w, h, s))
    torch.cuda.synchronize()
    sp = (steps & 0xffff).cpu().numpy(); ss = (steps >> 16).cpu().numpy() & 0xffff
    hit = rt.hit.cpu().numpy()
    th, tw = h // 8, w // 8
    tp = sp[:th * 8, :tw * 8].reshape(th, 8, tw, 8)
    ts = ss[:th * 8, :tw * 8].reshape(th, 8, tw, 8)
    thit = hit[:th * 8, :tw * 8].reshape(th, 8, tw, 8)
    mx_p, mean_p = tp.max(axis=(1, 3)), tp.mean(axis=(1, 3))
    mx_s, mean_s = ts.max(axis=(1, 3)), ts.mean(axis=(1, 3))
    anyhit = thit.max(axis=(1, 3))
    wave_evals = mx_p + mx_s + 7 * anyhit                       # scene evaluations a tile's waves execute
    lane_evals = (tp.sum(axis=(1, 3)) + ts.sum(axis=(1, 3)) + 7 * thit.sum(axis=(1, 3))) / 64.0
    print("tiles %d, wave-evaluations per tile: mean %.1f, median %.1f, p90 %.1f, max %d" %
          (wave_evals.size, wave_evals.mean(), np.median(wave_evals), np.percentile(wave_evals, 90), wave_evals.max()))
    print("lane utilisation (lane evaluations / wave evaluations): %.3f; primary only %.3f; shadow only %.3f" %
          (lane_evals.sum() / wave_evals.sum(), tp.sum() / 64.0 / mx_p.sum(), ts.sum() / 64.0 / max(mx_s.sum(), 1)))
    print("share of wave evaluations: primary %.3f shadow %.3f taps+colour %.3f" %
          (mx_p.sum() / wave_evals.sum(), mx_s.sum() / wave_evals.sum(), 7 * anyhit.sum() / wave_evals.sum()))
    print("hit tiles %d of %d; hit pixels %d" % (anyhit.sum(), anyhit.size, hit.sum()))
