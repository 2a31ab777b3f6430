#!/usr/bin/env python3
"""Developer tool: times rm_blit_device of several builds of librmhip.so on the default scene, interleaved, using only
the entry points every build has.   python scripts/dev_ab_lib.py W H lib1.so lib2.so ..."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raymarching_amd import scenes                      # noqa: E402
from raymarching_amd.scripts import get_frustum_corners  # noqa: E402


def load(path):
    l = C.CDLL(path)
    l.rm_create.restype = C.c_void_p
    l.rm_create.argtypes = [C.c_int]
    l.rm_destroy.argtypes = [C.c_void_p]
    l.rm_set_matrix.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float)]
    l.rm_set_float.argtypes = [C.c_void_p, C.c_char_p, C.c_float]
    l.rm_set_kernel.argtypes = [C.c_void_p, C.c_int]
    l.rm_blit_device.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
    return l


def main():
    w, h = int(sys.argv[1]), int(sys.argv[2])
    libs = [a for a in sys.argv[3:] if "=" not in a]
    opts = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
    steps = int(opts.get("steps", 300))
    cam = scenes.default_camera(w, h)
    fc = np.ascontiguousarray(get_frustum_corners(cam).m, dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    color = torch.empty((h, w, 4), dtype=torch.uint8, device="cuda")
    depth = torch.empty((h, w), dtype=torch.float32, device="cuda")
    hit = torch.empty((h, w), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    src = sd = None                                     # tex=1: bind a source image and a depth texture (all at the far plane)
    if opts.get("tex") == "1":
        tsrc = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda")
        tsd = torch.full((h, w), 1000.0, dtype=torch.float32, device="cuda")
        src, sd = tsrc.data_ptr(), tsd.data_ptr()
    ctxs = []
    for p in libs:
        l = load(p)
        c = l.rm_create(0)
        l.rm_set_matrix(c, b"_FrustrumCorners", fp(fc))
        if "kernel" in opts:
            l.rm_set_kernel(c, int(opts["kernel"]))
        for k, v in opts.items():
            if k.startswith("_"):
                l.rm_set_float(c, k.encode(), float(v))
        ctxs.append((p, l, c))
    for rnd in range(3):
        for p, l, c in ctxs:
            for _ in range(30):
                assert l.rm_blit_device(c, src, sd, color.data_ptr(), depth.data_ptr(), hit.data_ptr(), w, h, stream) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                l.rm_blit_device(c, src, sd, color.data_ptr(), depth.data_ptr(), hit.data_ptr(), w, h, stream)
            e1.record()
            torch.cuda.synchronize()
            print(f"round {rnd} {os.path.basename(p):28s} {w}x{h}: {e0.elapsed_time(e1) / steps * 1e3:8.1f} us/frame", flush=True)


if __name__ == "__main__":
    main()
