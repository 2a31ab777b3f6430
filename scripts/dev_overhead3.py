"""Developer tool: the floor of a 4K frame (_MaxSteps 0: no marching) by output planes, tile order and kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr
w, h = 3840, 2160
dev = "cuda:0"
c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
s = torch.cuda.Stream()
def run(tag, steps=0, planes=(1, 1, 1), floats=None, kernel=0):
    with Material() as m:
        Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()).bind()
        m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        m.SetFloat("_MaxSteps", steps)
        m.set_kernel(kernel)
        for k, v in (floats or {}).items(): m.SetFloat(k, v)
        f = lambda: m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c) if planes[0] else None, _ptr(d) if planes[1] else None,
                                                _ptr(hm) if planes[2] else None, w, h, s.cuda_stream))
        t = time.perf_counter()
        while time.perf_counter() - t < 0.05: f(); torch.cuda.synchronize()
        for _ in range(5): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): f()
        torch.cuda.synchronize()
        print("%-60s %.1f us" % (tag, (time.perf_counter() - t0) / 100 * 1e6))
run("steps 0, all planes (plain shape)")
run("steps 0, all static", floats={"_RmStaticFraction": 1.0})
run("steps 0, all static, no planes", planes=(0, 0, 0), floats={"_RmStaticFraction": 1.0})
run("steps 0, all static, no planes, no history", planes=(0, 0, 0), floats={"_RmStaticFraction": 1.0, "_RmTileFeedback": 0.0})
run("steps 0, all static, no planes, image order, no history", planes=(0, 0, 0), floats={"_RmStaticFraction": 1.0, "_RmTileFeedback": 0.0, "_RmScatterTiles": 0.0})
run("steps 0, all static, colour only", planes=(1, 0, 0), floats={"_RmStaticFraction": 1.0})
run("steps 0, default claims, no planes", planes=(0, 0, 0))
run("steps 1, all static, no planes", steps=1, planes=(0, 0, 0), floats={"_RmStaticFraction": 1.0})
run("steps 2, all static, no planes", steps=2, planes=(0, 0, 0), floats={"_RmStaticFraction": 1.0})
run("steps 4, all static, no planes", steps=4, planes=(0, 0, 0), floats={"_RmStaticFraction": 1.0})
