"""Developer tool: the shared-tile table kernel with and without exact culling (_RmTableSkip) on stress tables of several
sizes: bytes of the two frames compared, counters compared, time per frame.

    python scripts/dev_skip.py [n_prims ...]      (default 256 128 100 64 48 33)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                            # noqa: E402
from raymarching_amd import Material, Raymarching, RenderTexture, scenes   # noqa: E402
from raymarching_amd.material import _ptr, current_stream_handle       # noqa: E402

w, h = int(os.environ.get("RM_W", "1920")), int(os.environ.get("RM_H", "1080"))
sizes = [int(a) for a in sys.argv[1:]] or [256, 128, 100, 64, 48, 33]


def run(n, skip, frames):
    with Material() as mat:
        rm = Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms())
        rm.bind()
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        for k, v in scenes.STRESS_FLOATS.items():
            mat.SetFloat(k, v)
        mat.SetFloat("_RmTableSkip", 1.0 if skip else 0.0)
        mat.SetPrimitives(scenes.csg_random_primitives(n, seed=0x5EED0000 + n) if os.environ.get("RM_CSG") else
                          scenes.stress_primitives(n, seed=0x5EED0000 + n))
        rt = RenderTexture(w, h, device="cuda:0")
        st = current_stream_handle("cuda:0")

        def blit():
            mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, st))
        blit()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            blit()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / frames
        return ms, mat.stats(), rt.color.clone(), rt.depth.clone(), rt.hit.clone()


for n in sizes:
    frames = 3 if n >= 128 else 6
    a = run(n, False, frames)
    b = run(n, True, frames)
    same = all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(a[2:], b[2:])) and a[1] == b[1]
    print(f"n={n:4d} {w}x{h}: full walk {a[0]:8.3f} ms, exact culling {b[0]:8.3f} ms ({a[0] / b[0]:.2f}x), frames and counters "
          f"{'identical' if same else 'DIFFER'}  {b[1]}", flush=True)
