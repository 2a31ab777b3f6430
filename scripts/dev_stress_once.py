"""Developer tool: a few frames of the 256-primitive stress scene (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = int(os.environ.get("RM_W", "1920")), int(os.environ.get("RM_H", "1080"))
with Material() as mat:
    mat.set_kernel(int(os.environ.get("RM_KERNEL", "1")))
    rm = Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()); rm.bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    for k, v in scenes.STRESS_FLOATS.items(): mat.SetFloat(k, v)
    mat.SetPrimitives(scenes.stress_primitives(256))
    rt = RenderTexture(w, h, device="cuda:0")
    for _ in range(3):
        mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
    torch.cuda.synchronize()
    print(mat.stats())
