"""Developer tool: frame times of several scenes / sizes with the library named by RMHIP_LIB (A/B of end-game policies)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _fp, _ptr, current_stream_handle
import cases
dev = "cuda:0"
torch.cuda.set_device(0)
s = current_stream_handle(dev)
out = []
def run(label, w, h, setup, n=150):
    c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
    with Material(device=0) as m:
        setup(m, w, h)
        def frame():
            m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, s))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15: frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): frame()
        torch.cuda.synchronize()
        out.append("%s %.1f" % (label, (time.perf_counter() - t0) / n * 1e6))
def default(m, w, h):
    Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()).bind(); m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
def orbit(k):
    def f(m, w, h):
        Raymarching(scenes.orbit_camera(k, 64, w, h), m, scenes.default_transforms()).bind(); m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    return f
def transformed(m, w, h):
    Raymarching(scenes.default_camera(w, h), m, cases.transformed_transforms()).bind(); m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
for (w, h) in ((640, 360), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)):
    run(f"default_{w}x{h}", w, h, default, 150 if w < 7000 else 40)
for k in (5, 23, 40):
    run(f"orbit{k}_4k", 3840, 2160, orbit(k))
    run(f"orbit{k}_1080p", 1920, 1080, orbit(k))
run("transformed_4k", 3840, 2160, transformed)
run("transformed_1080p", 1920, 1080, transformed)
# a pose batch: 8 poses at 4K in one launch
m = Material(device=0); default(m, 3840, 2160)
n = 8
rm = Raymarching(scenes.default_camera(3840, 2160), m, scenes.default_transforms())
fcs = np.ascontiguousarray(np.stack([Raymarching(scenes.orbit_camera(k, 64, 3840, 2160), m, scenes.default_transforms()).GetFrustumCorners().m for k in range(n)]), dtype=np.float32)
cams = [scenes.orbit_camera(k, 64, 3840, 2160) for k in range(n)]
c2ws = np.ascontiguousarray(np.stack([cm.cameraToWorldMatrix.m for cm in cams]), dtype=np.float32)
pos = np.ascontiguousarray(np.stack([cm.transform.position for cm in cams]), dtype=np.float32)
c = torch.empty((n, 2160, 3840, 4), dtype=torch.uint8, device=dev); d = torch.empty((n, 2160, 3840), dtype=torch.float32, device=dev); hm = torch.empty((n, 2160, 3840), dtype=torch.uint8, device=dev)
def batch():
    m._ck(m._lib.rm_blit_poses_device(m.handle, n, _fp(fcs), _fp(c2ws), _fp(pos), _ptr(c), _ptr(d), _ptr(hm), 3840, 2160, s))
for _ in range(5): batch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): batch()
torch.cuda.synchronize()
out.append("poses8_4k_per_pose %.1f" % ((time.perf_counter() - t0) / 20 / n * 1e6))
m.close()
print(os.environ.get("RMHIP_LIB", "main")[-22:], " | ".join(out))
