"""Developer tool: a few 1920x1080 close-up frames of one scene() body (RM_BODY: 1 menger, 2 sierpinski, 3 mandelbulb, 4 mandelbrot,
5 terrain, 6 the composite), for rocprofv3 passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = int(os.environ.get("RM_W", "1920")), int(os.environ.get("RM_H", "1080"))
body = int(os.environ.get("RM_BODY", "1"))
dev = "cuda:0"
torch.cuda.set_device(0)
with Material(device=0) as m:
    ts = scenes.default_transforms()
    Raymarching(scenes.closeup_camera(ts[0].position, w, h), m, ts).bind()
    m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    m.SetFloat("_RmSceneBody", float(body))
    m.set_kernel(int(os.environ.get("RM_KERNEL", "0")))
    if "RM_THR" in os.environ:
        m.SetFloat("_RmHeavyTileThreshold", float(os.environ["RM_THR"]))
    c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
    for _ in range(int(os.environ.get("RM_FRAMES", "6"))):
        m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, current_stream_handle(dev)))
    torch.cuda.synchronize()
    print(m.stats())
