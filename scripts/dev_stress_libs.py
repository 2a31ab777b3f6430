"""Developer tool: the stress scene at 1080p with one kernel choice, once per library build given (each in a child process,
since a process loads one build):   python scripts/dev_stress_libs.py KERNEL lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:
    for lib in sys.argv[2:]:
        env = dict(os.environ, RMHIP_LIB=os.path.abspath(lib), RM_KERNEL=sys.argv[1])
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=170)
        print(os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from raymarching_amd import Material, Raymarching, RenderTexture, scenes
from raymarching_amd.material import _ptr, current_stream_handle
w, h = 1920, 1080
with Material() as mat:
    mat.set_kernel(int(os.environ.get("RM_KERNEL", "0")))
    Raymarching(scenes.stress_camera(w, h), mat, scenes.default_transforms()).bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    for k, v in scenes.STRESS_FLOATS.items():
        mat.SetFloat(k, v)
    mat.SetPrimitives(scenes.stress_primitives(256))
    rt = RenderTexture(w, h, device="cuda:0")
    step = lambda: mat._ck(mat._lib.rm_blit_device(mat.handle, None, None, _ptr(rt.color), _ptr(rt.depth), _ptr(rt.hit), w, h, current_stream_handle("cuda:0")))
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        step()
    e1.record(); torch.cuda.synchronize()
    import hashlib
    print(f"{e0.elapsed_time(e1) / 3:.2f} ms  sha {hashlib.sha256(rt.color.cpu().numpy().tobytes()).hexdigest()[:12]} {mat.stats()}")
