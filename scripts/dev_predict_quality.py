"""Developer tool: how well does the centre ray of an 8x8 tile predict the tile's cost (the maximum over its 64 rays of primary +
shadow evaluations + 7 for a hit)?  From the step-count plane of the default frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raymarching_amd import Material, Raymarching, scenes
from raymarching_amd.material import _ptr, current_stream_handle
dev = "cuda:0"
torch.cuda.set_device(0)
s = current_stream_handle(dev)
for w, h in ((3840, 2160), (1920, 1080)):
    c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
    steps = torch.zeros((h, w), dtype=torch.int32, device=dev)
    with Material(device=0) as m:
        Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()).bind()
        m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
        m._ck(m._lib.rm_blit_steps_device(m.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), _ptr(steps), w, h, s))
        torch.cuda.synchronize()
    st = steps.cpu().numpy(); hit = hm.cpu().numpy().astype(np.int64)
    cost = (st & 0xffff) + ((st >> 16) & 0xffff) + 7 * hit
    th, tw = h // 8, w // 8
    tiles = cost[:th * 8, :tw * 8].reshape(th, 8, tw, 8)
    true = tiles.max(axis=(1, 3))
    centre = tiles[:, 4, :, 4]
    quad = np.maximum.reduce([tiles[:, 2, :, 2], tiles[:, 2, :, 6], tiles[:, 6, :, 2], tiles[:, 6, :, 6]])
    print(f"{w}x{h}: {true.size} tiles; true cost >= 24: {(true >= 24).sum()}, >= 48: {(true >= 48).sum()}, max {true.max()}; sum of all costs {true.sum()}")
    for name, est in (("centre ray", centre), ("four rays (quadrant centres)", quad)):
        for thr in (8, 12, 16, 24):
            pred = np.minimum(est, 55) >= thr
            heavy = true >= 24
            very = true >= 48
            print(f"   {name:30s} threshold {thr:2d}: listed {pred.sum():6d}; of the {heavy.sum()} tiles >= 24 found {(pred & heavy).sum():5d}; "
                  f"of the {very.sum()} tiles >= 48 found {(pred & very).sum():5d}; cost of the tiles >= 48 missed: {true[very & ~pred].sum()}")
