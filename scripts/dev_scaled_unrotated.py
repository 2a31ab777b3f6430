"""Developer tool: frame time of the 4-slot scene with non-unit scales and no rotation (the reciprocal-exact scale variants) at 4K and 1080p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle
dev = "cuda:0"
torch.cuda.set_device(0)
s = current_stream_handle(dev)
SCL = [(1.3, 0.8, 1.1), (0.7, 1.9, 1.2), (0.9, 1.1, 1.05), (1.6, 0.6, 1.0)]
for w, h in ((3840, 2160), (1920, 1080)):
    c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
    for skip in (1.0, 0.0):
        with Material(device=0) as m:
            ts = scenes.default_transforms()
            for t, sc in zip(ts, SCL):
                t.lossyScale = np.asarray(sc, dtype=np.float32)
            Raymarching(scenes.default_camera(w, h), m, ts).bind()
            m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            m.SetFloat("_RmPairSkip", skip)
            def frame():
                m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, s))
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2: frame()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): frame()
            torch.cuda.synchronize()
            print("%dx%d _RmPairSkip %.0f: %.1f us per frame" % (w, h, skip, (time.perf_counter() - t0) / 200 * 1e6), m.stats(), flush=True)
