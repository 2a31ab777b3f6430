"""Developer tool: the cooperative table-scene kernel (four waves per 8x8 tile, a quarter of the table each) against the
one-wave-per-tile kernel: bit equality for several table sizes, then the time of the 256-primitive scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle


def run(kernel, n_prims, w, h, reps=3, seed=None):
    with Material() as mat:
        mat.set_kernel(kernel)
        Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()).bind()
        mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        for k, v in scenes.STRESS_FLOATS.items():
            mat.SetFloat(k, v)
        mat.SetPrimitives(scenes.stress_primitives(n_prims) if seed is None else scenes.stress_primitives(n_prims, seed=seed))
        rt = RenderTexture(w, h, device="cuda:0")

        def step():
            mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
        step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            step()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, mat.stats(), (rt.color.cpu().numpy(), rt.depth.cpu().numpy().view(np.uint32), rt.hit.cpu().numpy())


for n in (32, 33, 48, 64, 65, 96, 97, 100, 128, 200, 256, 257, 300, 512):
    a_ms, a_st, a = run(1, n, 333, 187, reps=1)
    b_ms, b_st, b = run(5, n, 333, 187, reps=1)
    same = all(np.array_equal(x, y) for x, y in zip(a, b)) and a_st == b_st
    print(f"n = {n}: tile {a_ms:.2f} ms, coop {b_ms:.2f} ms, identical {same}", flush=True)
    assert same, n
w, h = 1920, 1080
a_ms, a_st, a = run(1, 256, w, h)
b_ms, b_st, b = run(0, 256, w, h)
same = all(np.array_equal(x, y) for x, y in zip(a, b)) and a_st == b_st
print(f"stress256 1080p: tile {a_ms:.2f} ms, coop (auto) {b_ms:.2f} ms, {a_ms / b_ms:.2f}x, identical {same}  {b_st}", flush=True)
c_ms, c_st, c = run(3, 256, w, h)
same = all(np.array_equal(x, y) for x, y in zip(a, c)) and a_st == c_st
print(f"stress256 1080p: stream kernel with shared evaluations {c_ms:.2f} ms, {a_ms / c_ms:.2f}x the one-wave tile kernel, identical {same}", flush=True)
for n in (33, 100, 257):
    a_ms, a_st, a = run(1, n, 333, 187, reps=1)
    c_ms, c_st, c = run(3, n, 333, 187, reps=1)
    same = all(np.array_equal(x, y) for x, y in zip(a, c)) and a_st == c_st
    print(f"n = {n}: tile {a_ms:.2f} ms, stream-coop {c_ms:.2f} ms, identical {same}", flush=True)
    assert same, n
