"""Developer tool: the two depth passes from the light (ShadowMap.Render, ShadowMapDirectional) at 4K and 1080p on the tile kernel
and on the persistent kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raymarching_amd import Material, Raymarching, ShadowMap, ShadowMapDirectional, scenes
from raymarching_amd.material import _fp, _ptr, current_stream_handle
dev = "cuda:0"
torch.cuda.set_device(0)
stream = current_stream_handle(dev)
for w, h in ((3840, 2160), (1920, 1080)):
    smap = torch.empty((h, w), dtype=torch.float32, device=dev)
    ref = {}
    for kernel in (0, 1, 2):
        for mode in ("directional", "perspective"):
            with Material(device=0) as m:
                Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()).bind()
                m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
                m.set_kernel(kernel)
                if mode == "directional":
                    smd = ShadowMapDirectional(scenes.light_rectangle(), m)
                    smd.Update()
                    cn = np.ascontiguousarray(np.stack(smd.corners).astype(np.float32).reshape(12))
                    fw = np.ascontiguousarray(smd.transform.forward.astype(np.float32))
                    fn = lambda: m._ck(m._lib.rm_shadowmap_directional_device(m.handle, _fp(cn), _fp(fw), _ptr(smap), None, w, h, stream))
                else:
                    lc = scenes.light_camera(w, h)
                    smc = ShadowMap(lc, m)
                    m.SetMatrix("_FrustrumCorners", smc.GetFrustumCorners())
                    m.SetMatrix("_CameraInvViewMatrix", lc.cameraToWorldMatrix)
                    m.SetVector("_WorldSpaceCameraPos", lc.transform.position)
                    fn = lambda: m._ck(m._lib.rm_shadowmap_render_device(m.handle, _ptr(smap), None, w, h, stream))
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.2: fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100): fn()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 100 * 1e6
                if mode not in ref: ref[mode] = smap.clone()
                same = bool(torch.equal(ref[mode].view(torch.int32), smap.view(torch.int32)))
                print("%dx%d %-12s kernel %d: %7.1f us  %s  %s" % (w, h, mode, kernel, us, "same map" if same else "MAP DIFFERS", m.stats()), flush=True)
