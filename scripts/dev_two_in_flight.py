#!/usr/bin/env python3
"""Renders the default frame (1080p, or the size given after N) N times from one context on one stream, then 2N times from two contexts on two
streams (frames_in_flight 2).  Meant to run under `rocprofv3 --kernel-trace`: scripts/summarise_overlap.py reads the
dispatch intervals of the two phases from the trace.  Prints the wall-clock per frame of both phases."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from raymarching_amd import Material, Raymarching, scenes  # noqa: E402
from raymarching_amd.material import _ptr  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    dev = "cuda:0"
    torch.cuda.set_device(0)
    mats, frames = [], []
    for _ in range(2):
        m = Material(device=0)
        Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()).bind()
        m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        mats.append(m)
        frames.append((torch.empty((h, w), dtype=torch.int32, device=dev), torch.empty((h, w), dtype=torch.float32, device=dev),
                       torch.empty((h, w), dtype=torch.uint8, device=dev)))
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    def blit(i):
        m, (c, d, hm) = mats[i], frames[i]
        m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, streams[i].cuda_stream))

    for i in (0, 1):
        for _ in range(20):
            blit(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        blit(0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        blit(0)
        blit(1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    same = all(bool(torch.equal(a, b)) for a, b in zip(frames[0], frames[1]))
    print(json.dumps({"frames_single": n, "us_per_frame_single": (t1 - t0) / n * 1e6,
                      "frames_pair": 2 * n, "us_per_frame_two_in_flight": (t2 - t1) / (2 * n) * 1e6,
                      "frames_equal_between_contexts": same}))
    assert same


if __name__ == "__main__":
    main()
