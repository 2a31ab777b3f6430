"""Developer tool: the default frame with and without the frame-to-frame tile ordering (_RmTileFeedback), 4K and 1080p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle
dev = "cuda:0"
torch.cuda.set_device(0)
s = current_stream_handle(dev)
for w, h in ((3840, 2160), (1920, 1080)):
    c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
    for fb in (1.0, 0.0):
        with Material(device=0) as m:
            Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()).bind()
            m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            m.SetFloat("_RmTileFeedback", fb)
            def frame():
                m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, s))
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2: frame()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): frame()
            torch.cuda.synchronize()
            print("%dx%d _RmTileFeedback %.0f: %.1f us per frame" % (w, h, fb, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
