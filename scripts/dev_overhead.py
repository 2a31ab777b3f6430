"""Developer tool: per-tile overhead of the kernels = frame time with _MaxSteps 1 (one scene evaluation per pixel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
for (w, h) in ((1920, 1080), (3840, 2160)):
    for kernel in (1, 2):
        for steps in (1.0, 2.0, 400.0):
            with Material() as mat:
                mat.set_kernel(kernel)
                rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()); rm.bind()
                mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
                mat.SetFloat("_MaxSteps", steps)
                rt = RenderTexture(w, h, device="cuda:0")
                def step(): mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
                for _ in range(10): step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100): step()
                e1.record(); torch.cuda.synchronize()
                print(f"{w}x{h} kernel {kernel} max-steps {steps:.0f}: {e0.elapsed_time(e1)*10:.1f} us/frame")
