"""Developer tool: how uneven are the 8x8 tiles of a scene() body close-up (RM_BODY, 1080p)?  Frame time on the tile kernel and on the
persistent kernel, evaluations per tile (a wave executes the maximum over its 64 rays), and the longest tile against the
frame's mean load per wave slot."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = int(os.environ.get("RM_W", "1920")), int(os.environ.get("RM_H", "1080"))
body = int(os.environ.get("RM_BODY", "3"))
dev = "cuda:0"
torch.cuda.set_device(0)
s = current_stream_handle(dev)
c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
steps = torch.zeros((h, w), dtype=torch.int32, device=dev)
for kernel in (0, 1, 2):
    with Material(device=0) as m:
        ts = scenes.default_transforms()
        Raymarching(scenes.closeup_camera(ts[0].position, w, h), m, ts).bind()
        m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        m.SetFloat("_RmSceneBody", float(body))
        m.set_kernel(kernel)
        def frame():
            m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, s))
        for _ in range(3): frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): frame()
        torch.cuda.synchronize()
        print("kernel %d: %.3f ms per frame" % (kernel, (time.perf_counter() - t0) / 5 * 1e3), m.stats())
        if kernel == 0:
            m._ck(m._lib.rm_blit_steps_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), _ptr(steps), w, h, s))
            torch.cuda.synchronize()
sp = (steps & 0xffff).cpu().numpy(); ss = (steps >> 16).cpu().numpy() & 0xffff
hit = hm.cpu().numpy()
th, tw = h // 8, w // 8
tp = sp[:th * 8, :tw * 8].reshape(th, 8, tw, 8).max(axis=(1, 3))
tsd = ss[:th * 8, :tw * 8].reshape(th, 8, tw, 8).max(axis=(1, 3))
anyhit = hit[:th * 8, :tw * 8].reshape(th, 8, tw, 8).max(axis=(1, 3))
ev = tp + tsd + 7 * anyhit
print("tiles %d; wave evaluations per tile: mean %.1f median %.1f p99 %.1f max %d (primary %d + shadow %d)" %
      (ev.size, ev.mean(), np.median(ev), np.percentile(ev, 99), ev.max(), tp.reshape(-1)[ev.argmax()], tsd.reshape(-1)[ev.argmax()]))
print("mean load per wave slot (4096 slots): %.1f evaluations; the longest tile is %.2f of that" % (ev.sum() / 4096.0, ev.max() / (ev.sum() / 4096.0)))
iy, ix = np.unravel_index(ev.argmax(), ev.shape)
print("longest tile at tile row %d col %d (of %d x %d)" % (iy, ix, th, tw))
