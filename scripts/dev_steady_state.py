#!/usr/bin/env python3
"""How much of one frame's time is ramp and tail of its launch?  Renders the SAME pose n times: as n single launches
(rm_blit_device), as one launch of n identical poses (rm_blit_poses_device: one ramp, one tail, the same tiles n times),
and with two contexts on two streams.  All three produce the same frame(s); the per-frame times differ only by what a
launch costs beyond its arithmetic.

    python scripts/dev_steady_state.py [width height [n]]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raymarching_amd import Material, Raymarching, scenes  # noqa: E402
from raymarching_amd.material import _fp, _ptr  # noqa: E402


def timed(fn, reps):
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.05:
        fn()
        torch.cuda.synchronize()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 2 else 3840
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    dev = "cuda:0"
    torch.cuda.set_device(0)
    mats = []
    for _ in range(2):
        m = Material(device=0)
        rm = Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms())
        rm.bind()
        m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        mats.append(m)
    c = torch.empty((n, h, w, 4), dtype=torch.uint8, device=dev)
    d = torch.empty((n, h, w), dtype=torch.float32, device=dev)
    hm = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    c2, d2, hm2 = torch.empty_like(c[0]), torch.empty_like(d[0]), torch.empty_like(hm[0])
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    m = mats[0]
    fcs = np.ascontiguousarray(np.stack([rm.GetFrustumCorners().m] * n), dtype=np.float32)
    c2ws = np.ascontiguousarray(np.stack([rm.cam.cameraToWorldMatrix.m] * n), dtype=np.float32)
    pos = np.ascontiguousarray(np.stack([rm.cam.transform.position] * n), dtype=np.float32)
    s0 = streams[0].cuda_stream

    def single():
        m._ck(m._lib.rm_blit_device(m.handle, None, None, _ptr(c[0]), _ptr(d[0]), _ptr(hm[0]), w, h, s0))

    def batch():
        m._ck(m._lib.rm_blit_poses_device(m.handle, n, _fp(fcs), _fp(c2ws), _fp(pos), _ptr(c), _ptr(d), _ptr(hm), w, h, s0))

    def pair():
        single()
        m2 = mats[1]
        m2._ck(m2._lib.rm_blit_device(m2.handle, None, None, _ptr(c2), _ptr(d2), _ptr(hm2), w, h, streams[1].cuda_stream))

    t_single = timed(single, 100)
    t_batch = timed(batch, 12) / n
    t_pair = timed(pair, 100) / 2
    assert all(bool(torch.equal(c[i], c[0])) for i in range(1, n)) and bool(torch.equal(c2, c[0]))
    print(json.dumps({"size": f"{w}x{h}", "n": n, "us_per_frame_single_launches": t_single * 1e6,
                      "us_per_frame_in_one_launch_of_n": t_batch * 1e6, "us_per_frame_two_in_flight": t_pair * 1e6,
                      "launch_overhead_us": (t_single - t_batch) * 1e6}))


if __name__ == "__main__":
    main()
