import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
def run(kernel, feedback=1.0, thr=24.0, n=200, frac=0.5, bpc=0, w=1920, h=1080):
    mat = Material(); rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()); rm.bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    mat.set_kernel(kernel); mat.SetFloat("_RmTileFeedback", feedback); mat.SetFloat("_RmHeavyTileThreshold", thr); mat.SetFloat("_RmStaticFraction", frac); mat.SetFloat("_RmBlocksPerCU", bpc)
    rt = RenderTexture(w, h, device="cuda:0")
    def step(): mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
    for _ in range(20): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    wps, ns = C.c_int(), C.c_int()
    mat._ck(mat._lib.rm_debug_counter_slots(mat.handle, None, 0, C.byref(wps), C.byref(ns)))
    print(f"{w}x{h} bpc {bpc} kernel {kernel} feedback {feedback} thr {thr} frac {frac}: {e0.elapsed_time(e1)/n*1000:.1f} us/frame, slots {ns.value}")
    mat.close()
for (w, h) in ((1920, 1080), (3840, 2160)):
    for bpc in (0, 1, 2, 3, 4):
        for fr in (0.5, 0.0):
            run(2, bpc=bpc, frac=fr, w=w, h=h)
