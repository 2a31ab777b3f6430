import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from oracle import rm_oracle as O
for (W,H) in [(640,360),(1920,1080)]:
    mat = Material()
    rm = Raymarching(scenes.default_camera(W,H), mat, scenes.default_transforms())
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    rt = RenderTexture(W,H,device="cuda:0")
    rm.OnRenderImage(None, rt)
    torch.cuda.synchronize()
    st = mat.stats()
    uni = O.Uniforms(O.props_from_material(mat))
    ref = O.render(uni, W, H)
    rgba = rt.color.cpu().numpy(); depth = rt.depth.cpu().numpy(); hit = rt.hit.cpu().numpy()
    print(W,H,"gpu stats", st, "oracle", ref["stats"])
    print(" rgba equal:", np.array_equal(rgba, ref["rgba"]), "max diff", np.abs(rgba.astype(int)-ref["rgba"].astype(int)).max(),
          " depth bit-equal:", np.array_equal(depth.view(np.uint32), ref["depth"].view(np.uint32)), " hit equal:", np.array_equal(hit, ref["hit"]))
    # host path
    rt2 = RenderTexture(W,H)
    rm.OnRenderImage(None, rt2)
    print(" host path equal:", np.array_equal(rt2.color, rgba), np.array_equal(rt2.depth, depth))
    # timing
    for _ in range(5): rm.OnRenderImage(None, rt)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    N=50
    for _ in range(N): rm.OnRenderImage(None, rt)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/N
    flops = 292*st["sum_steps_p"] + 295*st["sum_steps_s"] + 2152*st["n_hit"]
    print(f" {ms*1000:.1f} us/frame  {(st['n_pixels']+st['n_hit'])/ms/1e3:.1f} Mrays/s  {flops/ms/1e9:.2f} TFLOP/s algorithmic")
