"""Developer tool: a few 3840x2160 frames whose shadow term is a shadow-map lookup (depth pass from the light + frame, directional
map unless RM_MODE=1), for rocprofv3 passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raymarching_amd import Material, Raymarching, ShadowMap, ShadowMapDirectional, scenes
from raymarching_amd.material import _fp, _ptr, current_stream_handle
from raymarching_amd.unity import Matrix4x4
w, h = int(os.environ.get("RM_W", "3840")), int(os.environ.get("RM_H", "2160"))
mode = int(os.environ.get("RM_MODE", "2"))
dev = "cuda:0"
torch.cuda.set_device(0)
stream = current_stream_handle(dev)


def material():
    m = Material(device=0)
    Raymarching(scenes.default_camera(w, h), m, scenes.default_transforms()).bind()
    m.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    return m


mf, msm = material(), material()
smap = torch.empty((h, w), dtype=torch.float32, device=dev)
c = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); d = torch.empty((h, w), dtype=torch.float32, device=dev); hm = torch.empty((h, w), dtype=torch.uint8, device=dev)
if mode == 2:
    smd = ShadowMapDirectional(scenes.light_rectangle(), msm)
    smd.Update()
    cn = np.ascontiguousarray(np.stack(smd.corners).astype(np.float32).reshape(12))
    fw = np.ascontiguousarray(smd.transform.forward.astype(np.float32))
    kk = Matrix4x4.identity()
    for i in range(4):
        kk.SetRow(i, smd.corners[i])
    mf.SetMatrix("_RmShadowMapCorners", kk)
    mf.SetVector("_RmShadowMapForward", smd.transform.forward)
    depth_pass = lambda: msm._ck(msm._lib.rm_shadowmap_directional_device(msm.handle, _fp(cn), _fp(fw), _ptr(smap), None, w, h, stream))
else:
    lc = scenes.light_camera(w, h)
    smc = ShadowMap(lc, msm)
    msm.SetMatrix("_FrustrumCorners", smc.GetFrustumCorners())
    msm.SetMatrix("_CameraInvViewMatrix", lc.cameraToWorldMatrix)
    msm.SetVector("_WorldSpaceCameraPos", lc.transform.position)
    mf.SetMatrix("_ShadowMapVPMatrix", lc.projectionMatrix * lc.worldToCameraMatrix)
    depth_pass = lambda: msm._ck(msm._lib.rm_shadowmap_render_device(msm.handle, _ptr(smap), None, w, h, stream))
mf.SetTexture("_ShadowMap", smap)
mf.SetFloat("_RmShadowMode", float(mode))
for _ in range(int(os.environ.get("RM_FRAMES", "40"))):
    depth_pass()
    mf._ck(mf._lib.rm_blit_device(mf.handle, None, None, _ptr(c), _ptr(d), _ptr(hm), w, h, stream))
torch.cuda.synchronize()
print(mf.stats())
mf.close(); msm.close()
