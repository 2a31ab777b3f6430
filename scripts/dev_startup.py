"""Developer tool: fixed cost of a launch = frame time of tiny frames, per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
for (w, h) in ((8, 8), (64, 64), (256, 256), (640, 360)):
    for kernel in (1, 2):
        with Material() as mat:
            mat.set_kernel(kernel)
            rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()); rm.bind()
            mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            mat.SetFloat("_MaxSteps", 1.0)
            rt = RenderTexture(w, h, device="cuda:0")
            def step(): mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
            for _ in range(20): step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(500): step()
            e1.record(); torch.cuda.synchronize()
            print(f"{w}x{h} kernel {kernel} one step per pixel: {e0.elapsed_time(e1)*2:.2f} us/frame")
