"""Developer tool: the persistent kernel on a scene() body close-up (RM_BODY, 1080p) under other heavy-tile thresholds and static
shares, against the tile kernel."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = int(os.environ.get("RM_W", "1920")), int(os.environ.get("RM_H", "1080"))
body = int(os.environ.get("RM_BODY", "3"))
dev = "cuda:0"
torch.cuda.set_device(0)
s = current_stream_handle(dev)
c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
ref = None


def run(kernel, floats):
    global ref
    with Material(device=0) as m:
        ts = scenes.default_transforms()
        Raymarching(scenes.closeup_camera(ts[0].position, w, h), m, ts).bind()
        m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        m.SetFloat("_RmSceneBody", float(body))
        for k, v in floats.items():
            m.SetFloat(k, v)
        m.set_kernel(kernel)
        def frame():
            m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, s))
        for _ in range(3): frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = int(os.environ.get("RM_FRAMES", "4"))
        for _ in range(n): frame()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        if ref is None:
            ref = c.clone()
        same = bool(torch.equal(ref, c))
        print("kernel %d %-60s %8.3f ms  %s" % (kernel, floats, ms, "same frame" if same else "FRAME DIFFERS"), flush=True)


run(1, {})
for thr, frac in itertools.product((24.0, 64.0, 128.0, 250.0), (-1.0, 0.0, 0.3)):
    run(2, {"_RmHeavyTileThreshold": thr, "_RmStaticFraction": frac})
run(2, {"_RmTileFeedback": 0.0, "_RmStaticFraction": 0.0})
