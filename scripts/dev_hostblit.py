"""Developer probe: the host-pointer rm_blit (PCIe copies included) next to the device-pointer call, 1080p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raymarching_amd import Graphics, Material, Raymarching, RenderTexture, scenes

w, h = 1920, 1080
mat = Material(device=0)
rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms()); rm.bind()
mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
rt = RenderTexture(w, h)                       # numpy targets on the host
for _ in range(5):
    Graphics.Blit(None, rt, mat)
n = 50
t0 = time.perf_counter()
for _ in range(n):
    Graphics.Blit(None, rt, mat)
dt = (time.perf_counter() - t0) / n
st = mat.stats()
rays = st["n_pixels"] + st["n_hit"]
print("host-pointer rm_blit 1080p: %.3f ms per frame = %.0f Mrays/s (synchronous, %.1f MB device-to-host per frame)"
      % (dt * 1e3, rays / dt / 1e6, w * h * 9 / 1e6))

# the same into page-locked buffers from rm_host_alloc
import ctypes as C
lib = mat._lib
n = w * h
pc, pd, ph = lib.rm_host_alloc(mat.handle, n * 4), lib.rm_host_alloc(mat.handle, n * 4), lib.rm_host_alloc(mat.handle, n)
assert pc and pd and ph
for _ in range(5):
    mat._ck(lib.rm_blit(mat.handle, None, None, pc, pd, ph, w, h))
reps = 50
t0 = time.perf_counter()
for _ in range(reps):
    mat._ck(lib.rm_blit(mat.handle, None, None, pc, pd, ph, w, h))
dt = (time.perf_counter() - t0) / reps
got = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint8)), shape=(h, w, 4))
assert np.array_equal(got, rt.color), "pinned-buffer frame differs"
print("host-pointer rm_blit 1080p into rm_host_alloc memory: %.3f ms per frame = %.0f Mrays/s" % (dt * 1e3, rays / dt / 1e6))
for p_ in (pc, pd, ph):
    lib.rm_host_free(mat.handle, p_)
mat.close()
