"""Developer tool: the 4-slot scene with rotations and non-unit scales on its own kernels (persistent queue, one wave per
tile) and the same four primitives as a table on the shared-tile kernel (one primitive per wave); 1080p, bit comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = 1920, 1080
out = {}
for tag, pow2, ident in (("rotated + scaled", False, False), ("untransformed", False, True)):
    ts = scenes.default_transforms() if ident else cases.transformed_transforms(pow2)
    table = [dict(p, pos=tuple(float(v) for v in t.position), quat=tuple(float(v) for v in t.rotation), scale=tuple(float(v) for v in t.lossyScale))
             for p, t in zip(cases.BY_NAME["prims4_as_table_320x180"].prims, ts)]
    for mode, kernel in (("slots, queue kernel", 0), ("slots, tile kernel", 1), ("table, shared tile", 5), ("table, one wave per tile", 1)):
        with Material() as mat:
            Raymarching(scenes.default_camera(w, h), mat, ts).bind()
            mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
            if mode.startswith("table"):
                mat.SetPrimitives(table)
            mat.set_kernel(kernel)
            rt = RenderTexture(w, h, device="cuda:0")
            step = lambda: mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                step()
            e1.record(); torch.cuda.synchronize()
            img = (rt.color.cpu().numpy().copy(), rt.depth.cpu().numpy().view(np.uint32).copy())
            ref = out.setdefault(tag, img)
            same = np.array_equal(ref[0], img[0]) and np.array_equal(ref[1], img[1])
            print(f"{tag}: {mode}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us, same bits as the first: {same}", flush=True)
