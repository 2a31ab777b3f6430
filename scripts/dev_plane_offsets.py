#!/usr/bin/env python3
"""Developer tool: the 4K default frame with its planes (colour, depth, hit and optionally source image, depth texture) placed at
different offsets from a 2 MiB boundary: planes that all start on such a boundary put one pixel's bytes of every plane on the same
HBM channel."""
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
    w, h = int(sys.argv[1]), int(sys.argv[2])
    kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lib = native.lib()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    n = w * h
    slot = (n * 4 + (8 << 20) + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    pool = torch.empty(5 * slot + (2 << 20), dtype=torch.uint8, device=dev)
    base = (-pool.data_ptr()) % (2 << 20)
    pool[:] = 0
    pool[base + 4 * slot: base + 5 * slot].view(torch.float32)[:] = 1000.0
    for tex in (False, True):
        for step_off in (0, 256, 1024, 4096, 4096 + 256, 8192, 65536, 65536 + 4096, 1 << 20):
            p = [pool.data_ptr() + base + i * slot + i * step_off for i in range(5)]
            m = default_material(w, h)
            if kernel:
                m.set_kernel(kernel)

            def step(_i):
                assert lib.rm_blit_device(m.handle, p[3] if tex else None, p[4] if tex else None, p[0], p[1], p[2], w, h, stream) == 0
            res = []
            for _ in range(3):
                el, ms = bench.timed(torch, step, 100, 20)
                res.append(ms * 1e3)
            print(f"tex {int(tex)} plane i at i*{step_off:8d} B past a 2 MiB boundary: " + " ".join(f"{r:7.1f}" for r in res) + " us", flush=True)
            m.close()


if __name__ == "__main__":
    main()
