"""Counterparts of the reference's three MonoBehaviours, method for method.  They
compute the same matrices in float32 and hand them to a ``Material`` exactly
where the C# scripts call ``mat.SetMatrix`` / ``Graphics.Blit``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np

from .material import Graphics, Material, RenderTexture, _fp, _ptr, current_stream_handle
from .unity import DEG2RAD, Camera, F, Matrix4x4, Transform, quat_rotate, vec3


def get_frustum_corners(cam: Camera) -> Matrix4x4:
    """GetFrustumCorners() (Assets/Raymarching.cs:32-56, duplicated at
    Assets/ShadowMap.cs:24-48): rows topLeft, topRight, bottomRight, bottomLeft of
    un-normalised view-space corner rays with z = -1."""
    cam_fov = F(cam.fieldOfView)
    cam_aspect = F(cam.aspect)
    fov_w_half = cam_fov * F(0.5)
    tan_fov = F(math.tan(float(fov_w_half * DEG2RAD)))          # Mathf.Tan
    to_right = vec3(1, 0, 0) * tan_fov * cam_aspect
    to_top = vec3(0, 1, 0) * tan_fov
    fwd = vec3(0, 0, 1)
    m = Matrix4x4.identity()
    m.SetRow(0, -fwd - to_right + to_top)
    m.SetRow(1, -fwd + to_right + to_top)
    m.SetRow(2, -fwd + to_right - to_top)
    m.SetRow(3, -fwd - to_right - to_top)
    return m


class Raymarching:
    """Assets/Raymarching.cs: binds five matrices and blits (OnRenderImage, :85-95)."""

    def __init__(self, cam: Camera, mat: Material, transforms: Sequence[Optional[Transform]], shadowMap=None):
        self.cam = cam
        self.mat = mat
        self.transforms: List[Optional[Transform]] = list(transforms)
        self.shadowMap = shadowMap            # a ShadowMap or ShadowMapDirectional; None = the script as it stands

    def GetFrustumCorners(self) -> Matrix4x4:
        return get_frustum_corners(self.cam)

    def GetPositions(self) -> Matrix4x4:                       # Raymarching.cs:58-65
        m = Matrix4x4.identity()
        for i in range(min(4, len(self.transforms))):
            if self.transforms[i] is not None:
                m.SetRow(i, self.transforms[i].position)
        return m

    def GetRotations(self) -> Matrix4x4:                       # Raymarching.cs:67-74
        m = Matrix4x4.identity()
        for i in range(min(4, len(self.transforms))):
            if self.transforms[i] is not None:
                m.SetRow(i, self.transforms[i].rotation)
        return m

    def GetScales(self) -> Matrix4x4:                          # Raymarching.cs:76-83
        m = Matrix4x4.identity()
        for i in range(min(4, len(self.transforms))):
            if self.transforms[i] is not None:
                m.SetRow(i, self.transforms[i].lossyScale)
        return m

    def bind(self) -> None:
        """The SetMatrix block of OnRenderImage (Raymarching.cs:86-90) plus the Unity
        built-ins the engine would have bound for this camera."""
        self.mat.SetMatrix("_FrustrumCorners", self.GetFrustumCorners())
        self.mat.SetMatrix("_CameraInvViewMatrix", self.cam.cameraToWorldMatrix)
        self.mat.SetMatrix("_Positions", self.GetPositions())
        self.mat.SetMatrix("_Rotations", self.GetRotations())
        self.mat.SetMatrix("_Scales", self.GetScales())
        self.mat.SetVector("_WorldSpaceCameraPos", self.cam.transform.position)
        near, far = F(self.cam.nearClipPlane), F(self.cam.farClipPlane)
        self.mat.SetVector("_ProjectionParams", [1.0, near, far, F(1) / far])
        self.mat.SetVector("_BackgroundColor", list(self.cam.backgroundColor[:3]) + [1.0])

    def bind_shadow_map(self) -> None:
        """The three lines OnRenderImage keeps commented out (Raymarching.cs:91-93): render the shadow map, bind it as
        `_ShadowMap`, bind the matrix that takes a world position into it.  For a ShadowMap camera that is
        `_ShadowMapVPMatrix` = projection * worldToCamera (the commented line writes the factors in the other order, which
        is not a transform of positions); for a ShadowMapDirectional rectangle its corners and forward."""
        sm = self.shadowMap
        sm.Render()
        self.mat.SetTexture("_ShadowMap", sm.shadowMap)
        if isinstance(sm, ShadowMap):
            self.mat.SetMatrix("_ShadowMapVPMatrix", sm.cam.projectionMatrix * sm.cam.worldToCameraMatrix)
            self.mat.SetFloat("_RmShadowMode", 1.0)
        else:
            k = Matrix4x4.identity()
            for i in range(4):
                k.SetRow(i, sm.corners[i])
            self.mat.SetMatrix("_RmShadowMapCorners", k)
            self.mat.SetVector("_RmShadowMapForward", sm.transform.forward)
            self.mat.SetFloat("_RmShadowMode", 2.0)

    def OnRenderImage(self, src: Optional[RenderTexture], dest: RenderTexture, scene_depth=None) -> None:
        self.bind()
        if self.shadowMap is not None:
            self.bind_shadow_map()
            self.bind()                       # the shadow pass may share this material: put the main camera back
        Graphics.Blit(src, dest, self.mat, scene_depth=scene_depth)


class ShadowMap:
    """Assets/ShadowMap.cs: Render() binds the shadow camera's two matrices and blits
    into ``shadowMap`` (:50-56).  The pass writes raymarch() depth (see
    rm_shadowmap_render in include/rmhip.h)."""

    def __init__(self, cam: Camera, shadowMapMaterial: Material, shadowMap: Optional[RenderTexture] = None):
        self.cam = cam
        self.shadowMapMaterial = shadowMapMaterial
        self.shadowMap = shadowMap

    def GetFrustumCorners(self) -> Matrix4x4:
        return get_frustum_corners(self.cam)

    def Render(self) -> None:
        if self.shadowMap is None:
            self.shadowMap = RenderTexture(100, 100)            # the 100x100 temporary of ShadowMap.cs:51
        m = self.shadowMapMaterial
        m.SetMatrix("_FrustrumCorners", self.GetFrustumCorners())
        m.SetMatrix("_CameraInvViewMatrix", self.cam.cameraToWorldMatrix)
        m.SetVector("_WorldSpaceCameraPos", self.cam.transform.position)
        rt = self.shadowMap
        if rt.device is None:
            m._ck(m._lib.rm_shadowmap_render(m.handle, _ptr(rt.depth), rt.width, rt.height))
        else:
            m._ck(m._lib.rm_shadowmap_render_device(m.handle, _ptr(rt.depth), _ptr(rt.hit), rt.width, rt.height,
                                                    current_stream_handle(rt.device)))


class ShadowMapDirectional:
    """Assets/ShadowMapDirectional.cs: Update() computes the four world-space corners
    of the oriented rectangle (:25-34); Render() marches orthographic rays from it."""

    initialCorners = (vec3(-0.5, 0.5, 0), vec3(0.5, 0.5, 0), vec3(0.5, -0.5, 0), vec3(-0.5, -0.5, 0))

    def __init__(self, transform: Transform, material: Material, shadowMap: Optional[RenderTexture] = None):
        self.transform = transform
        self.material = material
        self.shadowMap = shadowMap
        self.corners = [vec3(0, 0, 0) for _ in range(4)]

    def Update(self) -> None:
        t = self.transform
        for i in range(4):
            c = self.initialCorners[i].copy()
            c[0] *= t.lossyScale[0]
            c[1] *= t.lossyScale[1]
            c[2] *= t.lossyScale[2]
            c = quat_rotate(t.rotation, c)
            c = c + t.position
            self.corners[i] = c.astype(F)

    def Render(self) -> None:
        if self.shadowMap is None:
            raise ValueError("shadowMap render target not set")
        self.Update()
        m = self.material
        corners = np.ascontiguousarray(np.stack(self.corners).astype(F).reshape(12))
        fwd = np.ascontiguousarray(self.transform.forward.astype(F))
        rt = self.shadowMap
        if rt.device is None:
            m._ck(m._lib.rm_shadowmap_directional(m.handle, _fp(corners), _fp(fwd), _ptr(rt.depth), rt.width, rt.height))
        else:
            m._ck(m._lib.rm_shadowmap_directional_device(m.handle, _fp(corners), _fp(fwd), _ptr(rt.depth), _ptr(rt.hit),
                                                         rt.width, rt.height, current_stream_handle(rt.device)))
