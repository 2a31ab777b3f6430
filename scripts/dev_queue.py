"""Developer tool: sweeps the queue kernel's scheduling properties on the default scene."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle

def run(w, h, thr=24.0, frac=0.5, n=200):
    mat = Material(); rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()); rm.bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    mat.SetFloat("_RmHeavyTileThreshold", thr); mat.SetFloat("_RmStaticFraction", frac)
    rt = RenderTexture(w, h, device="cuda:0")
    def step(): mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
    for _ in range(20): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    print(f"{w}x{h} heavy-thr {thr} static {frac}: {e0.elapsed_time(e1)/n*1000:.1f} us/frame")
    mat.close()

for (w, h) in ((640, 360), (1920, 1080), (3840, 2160), (7680, 4320)):
    for thr in (24.0, 40.0):
        for fr in (-1.0, 0.25, 0.5):
            run(w, h, thr, fr, n=200 if w < 7000 else 40)
