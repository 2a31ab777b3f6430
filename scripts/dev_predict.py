"""Developer tool: frames WITHOUT a usable previous frame (the frame size alternates between h and h - 8 rows, so every launch
finds another tile count and starts from cleared queue state) with the tile-cost prediction pass on and off, against the same
frames rendered repeatedly (history).  Optional argument: _RmPredictThreshold."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle
dev = "cuda:0"
torch.cuda.set_device(0)
s = current_stream_handle(dev)
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
cap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for w, h in ((3840, 2160), (1920, 1080), (1280, 720)):
    c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
    ref = None
    for label, floats, alternate in (("history", {}, False), ("no history, prediction", {"_RmPredictThreshold": thr}, True),
                                     ("no history, no prediction", {"_RmPredictTiles": 0.0}, True)):
        with Material(device=0) as m:
            rm = Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()); rm.bind()
            m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            for k, v in floats.items():
                m.SetFloat(k, v)
            def frame(hh):
                m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, hh, s))
            def pair():
                if alternate:
                    frame(h - 8)
                frame(h)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2: pair()
            torch.cuda.synchronize()
            n = 100
            t0 = time.perf_counter()
            for _ in range(n): pair()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / n / (2 if alternate else 1) * 1e6
            if ref is None: ref = c.clone()
            print("%dx%d %-28s %7.1f us per frame  %s" % (w, h, label, us, "same frame" if bool(torch.equal(ref, c)) else "FRAME DIFFERS"), flush=True)
