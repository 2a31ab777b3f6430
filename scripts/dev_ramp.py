#!/usr/bin/env python3
"""How long does the clock take to settle?  Times K = 20 frames of the 4K default scene after an idle gap and different lead-ins:
the fenced lead-in of bench.timed (a synchronize every 4 frames), an unfenced one of the same length, longer ones, and K = 200."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle

w, h = 3840, 2160
dev = "cuda:0"
torch.cuda.set_device(0)
stream = current_stream_handle(dev)
c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)


def run(lead, fenced, K, W=5):
    time.sleep(0.3)                                         # idle gap
    m = Material(device=0)
    Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()).bind()
    m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    def fn():
        m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, stream))
    if lead > 0:
        if fenced:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < lead:
                for _ in range(4):
                    fn()
                torch.cuda.synchronize()
        else:
            for _ in range(int(lead / 0.00025)):
                fn()
            torch.cuda.synchronize()
    for _ in range(W):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    m.close()
    return e0.elapsed_time(e1) / K * 1e3


out = {}
for rep in range(3):
    for name, args in (("fenced_30ms_K20", (0.03, True, 20)), ("unfenced_30ms_K20", (0.03, False, 20)), ("unfenced_100ms_K20", (0.1, False, 20)),
                       ("unfenced_300ms_K20", (0.3, False, 20)), ("none_K20", (0.0, False, 20)), ("fenced_30ms_K200", (0.03, True, 200)),
                       ("fenced_30ms_K2000", (0.03, True, 2000))):
        out.setdefault(name, []).append(round(run(*args), 1))
print(json.dumps(out))
