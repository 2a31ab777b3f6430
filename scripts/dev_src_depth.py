#!/usr/bin/env python3
"""Developer tool: the 4K default frame with and without a source image and a depth texture bound, same context set-up as bench.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402
from raymarching_amd import Material, Raymarching, native, scenes   # noqa: E402


def default_material(ww, hh):
    m = Material(device=0)
    Raymarching(scenes.default_camera(ww, hh), m, scenes.default_transforms()).bind()
    m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    return m


def main():
    w, h = 3840, 2160
    lib = native.lib()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    color = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
    depth = torch.empty((h, w), dtype=torch.float32, device=dev)
    hit = torch.empty((h, w), dtype=torch.uint8, device=dev)
    srcs = {"random": torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev),
            "zeros": torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)}
    sds = {"1000": torch.full((h, w), 1000.0, dtype=torch.float32, device=dev),
           "1e9": torch.full((h, w), 1e9, dtype=torch.float32, device=dev)}
    cases = [("none", None, None)] + [(f"src {a} depth {b}", srcs[a], sds[b]) for a in srcs for b in sds]
    for warm in (10, 40):
        for name, s, d in cases:
            m = default_material(w, h)
            sp = s.data_ptr() if s is not None else None
            dp = d.data_ptr() if d is not None else None

            def step(_i):
                assert lib.rm_blit_device(m.handle, sp, dp, color.data_ptr(), depth.data_ptr(), hit.data_ptr(), w, h, stream) == 0
            el, ms = bench.timed(torch, step, 50, warm)
            st = m.stats()
            print(f"warmup {warm:3d} {name:28s} {ms * 1e3:7.1f} us  hits {st['n_hit']} primary {st['sum_steps_p']}", flush=True)
            m.close()


if __name__ == "__main__":
    main()
