"""Developer tool (needs RMHIP_LIB = a -DRM_TIMELINE build): when the waves of one 4K frame start and end, with the march
switched off (_MaxSteps 0) and on, all tiles dealt statically or with the default claims."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
w, h = 3840, 2160
def run(tag, steps, floats):
    with Material() as mat:
        rm = Raymarching(scenes.default_camera(w, h), mat, scenes.default_transforms())
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        mat.SetFloat("_MaxSteps", steps)
        for k, v in floats.items(): mat.SetFloat(k, v)
        rt = RenderTexture(w, h, device="cuda:0")
        for _ in range(20): rm.OnRenderImage(None, rt)
        torch.cuda.synchronize()
        wps, ns = C.c_int(), C.c_int()
        mat._ck(mat._lib.rm_debug_counter_slots(mat.handle, None, 0, C.byref(wps), C.byref(ns)))
        buf = np.zeros(wps.value * ns.value, dtype=np.uint32)
        mat._ck(mat._lib.rm_debug_counter_slots(mat.handle, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size, None, None))
        s = buf.reshape(ns.value, wps.value)
        assert wps.value == 16, "not a timeline build"
        t0 = s[:, 3].astype(np.uint64) | (s[:, 4].astype(np.uint64) << np.uint64(32))
        t1 = s[:, 5].astype(np.uint64) | (s[:, 6].astype(np.uint64) << np.uint64(32))
        base = t0.min()
        st, en = (t0 - base).astype(np.float64) / 100.0, (t1 - base).astype(np.float64) / 100.0
        nd = (s[:, 7] & 0xffff).astype(np.float64)
        print(f"{tag}: waves {ns.value}; span {en.max():.1f} us; wave start p50/p99/max {np.median(st):.1f}/{np.percentile(st, 99):.1f}/{st.max():.1f}; "
              f"end p10/p50/p90/max {np.percentile(en, 10):.1f}/{np.median(en):.1f}/{np.percentile(en, 90):.1f}/{en.max():.1f}; tiles per wave mean {nd.mean():.1f} max {nd.max():.0f}; "
              f"us per tile (own time / tiles) p50 {np.median((en - st) / np.maximum(nd, 1)):.2f}")
run("steps 0, all static", 0, {"_RmStaticFraction": 1.0})
run("steps 0, default", 0, {})
run("steps 400, default", 400, {})
