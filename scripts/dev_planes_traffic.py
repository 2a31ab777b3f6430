"""Developer tool (run under rocprofv3 --pmc WRITE_SIZE): the 4K default frame with or without the hit-mask plane
(RM_NO_MASK=1), 20 frames.  Shows which output plane the HBM write traffic beyond the algorithmic bytes belongs to."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                            # noqa: E402
from raymarching_amd import Material, Raymarching, RenderTexture, scenes   # noqa: E402
from raymarching_amd.material import _ptr, current_stream_handle       # noqa: E402

w, h = 3840, 2160
with Material() as mat:
    Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()).bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    rt = RenderTexture(w, h, device="cuda:0")
    hit = None if os.environ.get("RM_NO_MASK") else _ptr(rt.hit)
    for _ in range(20):
        mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), hit, w, h, current_stream_handle("cuda:0")))
    torch.cuda.synchronize()
