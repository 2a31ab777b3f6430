"""ctypes binding of the CPU oracle (oracle/librm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg, never by the raymarching_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librm_oracle.so")


class RmoPrim(C.Structure):
    _fields_ = [("type", C.c_int32), ("op", C.c_int32), ("pos", C.c_float * 3), ("quat", C.c_float * 4),
                ("scale", C.c_float * 3), ("color", C.c_float * 3), ("params", C.c_float * 4)]


class RmoUniforms(C.Structure):
    _fields_ = [("frustum_corners", C.c_float * 16), ("cam_inv_view", C.c_float * 16),
                ("positions", C.c_float * 16), ("rotations", C.c_float * 16), ("scales", C.c_float * 16),
                ("cam_pos", C.c_float * 3), ("light", C.c_float * 4),
                ("max_steps", C.c_float), ("max_dist", C.c_float), ("step_factor", C.c_float),
                ("contact_threshold", C.c_float), ("normal_sample_scale", C.c_float), ("far_plane", C.c_float),
                ("smooth_k", C.c_float), ("background", C.c_float * 4),
                ("n_prims", C.c_int32), ("prims", C.POINTER(RmoPrim)), ("shade_miss_pixels", C.c_int32),
                ("scene_body", C.c_int32), ("fractal_scale", C.c_float), ("fractal_rotation", C.c_float * 3),
                ("fractal_power", C.c_float), ("fx_step_heat", C.c_int32), ("fx_debug_view", C.c_int32),
                ("fx_volumetric", C.c_int32), ("fx_bounce", C.c_int32),
                ("fx_light_model", C.c_int32), ("fx_scatter", C.c_int32), ("light_position", C.c_float * 3),
                ("refractive_index", C.c_float),
                ("domain_op", C.c_int32), ("domain_period", C.c_float * 3), ("domain_size", C.c_float),
                ("shadow_mode", C.c_int32), ("shadow_map", C.POINTER(C.c_float)), ("shadow_w", C.c_int32), ("shadow_h", C.c_int32),
                ("shadow_vp", C.c_float * 16), ("shadow_corners", C.c_float * 12), ("shadow_forward", C.c_float * 3),
                ("shadow_bias", C.c_float), ("time", C.c_float * 4), ("animate_fractal", C.c_int32)]


class RmoStats(C.Structure):
    _fields_ = [("n_pixels", C.c_uint64), ("n_hit", C.c_uint64), ("sum_steps_p", C.c_uint64),
                ("sum_steps_s", C.c_uint64), ("sum_steps_s_all", C.c_uint64)]


_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        FP = C.POINTER(C.c_float)
        UP = C.POINTER(RmoUniforms)
        l.rmo_render.restype = C.c_int
        l.rmo_render.argtypes = [UP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(RmoStats)]
        l.rmo_shadowmap_perspective.restype = C.c_int
        l.rmo_shadowmap_perspective.argtypes = [UP, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                C.POINTER(RmoStats)]
        l.rmo_shadowmap_directional.restype = C.c_int
        l.rmo_shadowmap_directional.argtypes = [UP, FP, FP, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                C.POINTER(RmoStats)]
        l.rmo_sphere.restype = C.c_float
        l.rmo_sphere.argtypes = [FP, C.c_float]
        l.rmo_box.restype = C.c_float
        l.rmo_box.argtypes = [FP, FP]
        l.rmo_link.restype = C.c_float
        l.rmo_link.argtypes = [FP, C.c_float, C.c_float, C.c_float]
        l.rmo_cylinder.restype = C.c_float
        l.rmo_cylinder.argtypes = [FP, C.c_float, C.c_float]
        l.rmo_torus.restype = C.c_float
        l.rmo_torus.argtypes = [FP, C.c_float, C.c_float]
        for fn in (l.rmo_mod, l.rmo_shmod):
            fn.restype = C.c_float
            fn.argtypes = [C.c_float, C.c_float]
        for fn in (l.rmo_boxfold, l.rmo_spherefold):
            fn.restype = None
            fn.argtypes = [FP, C.c_float, FP]
        l.rmo_rotate_q.restype = None
        l.rmo_rotate_q.argtypes = [FP, FP, FP]
        l.rmo_remap.restype = C.c_float
        l.rmo_remap.argtypes = [C.c_float] * 5
        l.rmo_smin4.restype = None
        l.rmo_smin4.argtypes = [FP, FP, C.c_float, FP]
        for fn in (l.rmo_c_sin, l.rmo_c_cos, l.rmo_c_acos, l.rmo_c_asin, l.rmo_c_log):
            fn.restype = C.c_float
            fn.argtypes = [C.c_float]
        l.rmo_c_atan.restype = C.c_float
        l.rmo_c_atan.argtypes = [C.c_float]
        for fn in (l.rmo_rand, l.rmo_noise, l.rmo_terrain):
            fn.restype = C.c_float
            fn.argtypes = [FP]
        l.rmo_getbrightness.restype = C.c_float
        l.rmo_getbrightness.argtypes = [FP, FP, FP, C.c_int]
        l.rmo_getscatter.restype = C.c_float
        l.rmo_getscatter.argtypes = [FP, FP, C.c_float, FP]
        l.rmo_refract2.restype = None
        l.rmo_refract2.argtypes = [FP, FP, C.c_float, FP]
        for fn in (l.rmo_c_atan2, l.rmo_c_pow):
            fn.restype = C.c_float
            fn.argtypes = [C.c_float, C.c_float]
        for fn in (l.rmo_mandelbulb, l.rmo_mandelbrot):
            fn.restype = None
            fn.argtypes = [FP, C.c_float, FP]
        for fn in (l.rmo_menger, l.rmo_sierpinski):
            fn.restype = None
            fn.argtypes = [C.POINTER(RmoUniforms), FP, FP]
        l.rmo_smax4.restype = None
        l.rmo_smax4.argtypes = [FP, FP, C.c_float, FP]
        for fn in (l.rmo_smin1, l.rmo_smax1):
            fn.restype = C.c_float
            fn.argtypes = [C.c_float, C.c_float, C.c_float]
        l.rmo_sq.restype = C.c_float
        l.rmo_sq.argtypes = [C.c_float]
        l.rmo_sq3.restype = None
        l.rmo_sq3.argtypes = [FP, FP]
        l.rmo_luminance.restype = C.c_float
        l.rmo_luminance.argtypes = [FP]
        l.rmo_getnormal.restype = None
        l.rmo_getnormal.argtypes = [UP, FP, FP, FP]
        l.rmo_scene.restype = None
        l.rmo_scene.argtypes = [UP, FP, FP]
        l.rmo_pixel_ray.restype = None
        l.rmo_pixel_ray.argtypes = [UP, C.c_int, C.c_int, C.c_int, C.c_int, FP]
        l.rmo_vert_corner_index.restype = C.c_int
        l.rmo_vert_corner_index.argtypes = [C.c_float, C.c_float]
        l.rmo_raymarch.restype = C.c_int
        l.rmo_raymarch.argtypes = [UP, FP, FP, C.c_float, FP, C.POINTER(C.c_int)]
        l.rmo_online_cpus.restype = C.c_int
        l.rmo_online_cpus.argtypes = []
        _lib = l
    return _lib


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))


def fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Uniforms:
    """Owns an RmoUniforms and the primitive array it may point to."""

    def __init__(self, props: dict, prims: Optional[Sequence[dict]] = None, shade_miss_pixels: bool = False,
                 shadow_map: Optional[np.ndarray] = None):
        u = RmoUniforms()
        u.frustum_corners[:] = f32(props["_FrustrumCorners"]).tolist()
        u.cam_inv_view[:] = f32(props["_CameraInvViewMatrix"]).tolist()
        u.positions[:] = f32(props["_Positions"]).tolist()
        u.rotations[:] = f32(props["_Rotations"]).tolist()
        u.scales[:] = f32(props["_Scales"]).tolist()
        u.cam_pos[:] = f32(props["_WorldSpaceCameraPos"])[:3].tolist()
        u.light[:] = f32(props["_WorldSpaceLightPos0"])[:4].tolist()
        u.max_steps = props["_MaxSteps"]
        u.max_dist = props["_MaxDist"]
        u.step_factor = props["_StepFactor"]
        u.contact_threshold = props["_ContactThreshold"]
        u.normal_sample_scale = props["_NormalSampleScale"]
        u.far_plane = float(f32(props["_ProjectionParams"])[2])
        u.smooth_k = 0.5
        u.background[:] = f32(props["_BackgroundColor"])[:4].tolist()
        self._prims = None
        if prims:
            arr = (RmoPrim * len(prims))()
            for i, p in enumerate(prims):
                arr[i].type = int(p["type"])
                arr[i].op = int(p.get("op", 0))
                arr[i].pos[:] = [float(x) for x in p["pos"]]
                arr[i].quat[:] = [float(x) for x in p["quat"]]
                arr[i].scale[:] = [float(x) for x in p["scale"]]
                arr[i].color[:] = [float(x) for x in p["color"]]
                pr = list(p.get("params", ())) + [0.0] * 4
                arr[i].params[:] = [float(x) for x in pr[:4]]
            self._prims = arr
            u.n_prims = len(prims)
            u.prims = C.cast(arr, C.POINTER(RmoPrim))
        else:
            u.n_prims = 0
        u.shade_miss_pixels = 1 if shade_miss_pixels else 0
        u.scene_body = int(props.get("_RmSceneBody", 0))
        u.fractal_scale = props.get("_FractalScale", 2.0)
        u.fractal_power = props.get("_RmFractalPower", 0.0)
        u.fx_step_heat = int(props.get("_RmStepHeat", 0) != 0)
        u.fx_debug_view = int(props.get("_RmDebugView", 0))
        u.fx_volumetric = int(props.get("_RmVolumetricLight", 0) != 0)
        u.fx_bounce = int(props.get("_RmBounce", 0))
        u.fx_light_model = int(props.get("_RmLightModel", 0))
        u.fx_scatter = int(props.get("_RmScatter", 0) != 0)
        u.light_position[:] = f32(props.get("_LightPosition", np.zeros(4)))[:3].tolist()
        u.refractive_index = props.get("_RefractiveIndex", 1.0)
        u.domain_op = int(props.get("_RmDomainOp", 0))
        u.domain_period[:] = [float(x) for x in list(props.get("_RmDomainPeriod", (0.0, 0.0, 0.0, 0.0)))[:3]]
        u.domain_size = props.get("_RmDomainSize", 1.0)
        u.fractal_rotation[:] = [props.get("_FractalRotationX", 0.0), props.get("_FractalRotationY", 0.0),
                                 props.get("_FractalRotationZ", 0.0)]
        u.time[:] = f32(props.get("_Time", np.zeros(4))).tolist()
        u.animate_fractal = int(props.get("_RmAnimateFractal", 0) != 0)
        u.shadow_mode = int(props.get("_RmShadowMode", 0))
        u.shadow_bias = props.get("_RmShadowBias", 0.01)
        u.shadow_vp[:] = f32(props.get("_ShadowMapVPMatrix", np.eye(4))).tolist()
        u.shadow_corners[:] = f32(props.get("_RmShadowMapCorners", np.zeros(16)))[[0, 4, 8, 1, 5, 9, 2, 6, 10, 3, 7, 11]].tolist()
        u.shadow_forward[:] = f32(props.get("_RmShadowMapForward", np.zeros(4)))[:3].tolist()
        self._shadow_map = None
        if u.shadow_mode != 0:
            if shadow_map is None:
                raise ValueError("_RmShadowMode set but no shadow map given")
            self._shadow_map = np.ascontiguousarray(shadow_map, dtype=np.float32)
            u.shadow_map = self._shadow_map.ctypes.data_as(C.POINTER(C.c_float))
            u.shadow_h, u.shadow_w = self._shadow_map.shape
        self.u = u

    @property
    def ref(self):
        return C.byref(self.u)


PROP_MATRICES = ("_FrustrumCorners", "_CameraInvViewMatrix", "_Positions", "_Rotations", "_Scales", "_ShadowMapVPMatrix",
                 "_RmShadowMapCorners")
PROP_VECTORS = ("_WorldSpaceCameraPos", "_WorldSpaceLightPos0", "_ProjectionParams", "_BackgroundColor", "_RmDomainPeriod", "_RmShadowMapForward", "_LightPosition", "_Time")
PROP_FLOATS = ("_MaxSteps", "_MaxDist", "_StepFactor", "_ContactThreshold", "_NormalSampleScale",
               "_FractalScale", "_FractalRotationX", "_FractalRotationY", "_FractalRotationZ", "_RmSceneBody", "_RmFractalPower",
               "_RmStepHeat", "_RmDebugView", "_RmVolumetricLight", "_RmBounce", "_RmDomainOp", "_RmDomainSize", "_RmShadowMode", "_RmShadowBias", "_RmLightModel", "_RmScatter", "_RefractiveIndex", "_RmAnimateFractal")


def props_from_material(mat) -> dict:
    """Reads back, through the plugin's own getters, every property the march uses."""
    p = {}
    for n in PROP_MATRICES:
        p[n] = mat.GetMatrix(n).m.copy()
    for n in PROP_VECTORS:
        p[n] = mat.GetVector(n)
    for n in PROP_FLOATS:
        p[n] = mat.GetFloat(n)
    return p


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def render(uni: Uniforms, w: int, h: int, src: Optional[np.ndarray] = None, scene_depth: Optional[np.ndarray] = None,
           y0: int = 0, y1: Optional[int] = None, threads: int = 0, want_float: bool = False, want_steps: bool = False):
    """Returns dict(rgba[h,w,4] u8, depth[h,w] f32, hit[h,w] u8, stats, [rgbf], [steps_p, steps_s])."""
    y1 = h if y1 is None else y1
    rgba = np.zeros((h, w, 4), dtype=np.uint8)
    depth = np.zeros((h, w), dtype=np.float32)
    hit = np.zeros((h, w), dtype=np.uint8)
    rgbf = np.zeros((h, w, 3), dtype=np.float32) if want_float else None
    sp = np.zeros((h, w), dtype=np.uint16) if want_steps else None
    ss = np.zeros((h, w), dtype=np.uint16) if want_steps else None
    if src is not None:
        src = np.ascontiguousarray(src, dtype=np.uint8)
    if scene_depth is not None:
        scene_depth = np.ascontiguousarray(scene_depth, dtype=np.float32)
    st = RmoStats()
    rc = lib().rmo_render(uni.ref, _vp(src), _vp(scene_depth), _vp(rgba), _vp(depth), _vp(hit), _vp(rgbf), _vp(sp),
                          _vp(ss), w, h, y0, y1, threads, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"rmo_render failed: {rc}")
    out = {"rgba": rgba, "depth": depth, "hit": hit,
           "stats": {"n_pixels": st.n_pixels, "n_hit": st.n_hit, "sum_steps_p": st.sum_steps_p,
                     "sum_steps_s": st.sum_steps_s, "sum_steps_s_all": st.sum_steps_s_all}}
    if want_float:
        out["rgbf"] = rgbf
    if want_steps:
        out["steps_p"], out["steps_s"] = sp, ss
    return out


def shadowmap_perspective(uni: Uniforms, w: int, h: int, threads: int = 0):
    depth = np.zeros((h, w), dtype=np.float32)
    hit = np.zeros((h, w), dtype=np.uint8)
    st = RmoStats()
    rc = lib().rmo_shadowmap_perspective(uni.ref, _vp(depth), _vp(hit), w, h, threads, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"rmo_shadowmap_perspective failed: {rc}")
    return {"depth": depth, "hit": hit, "stats": {"n_hit": st.n_hit, "sum_steps_p": st.sum_steps_p}}


def shadowmap_directional(uni: Uniforms, corners, forward, w: int, h: int, threads: int = 0):
    depth = np.zeros((h, w), dtype=np.float32)
    hit = np.zeros((h, w), dtype=np.uint8)
    st = RmoStats()
    c, f = f32(corners), f32(forward)
    rc = lib().rmo_shadowmap_directional(uni.ref, fp(c), fp(f), _vp(depth), _vp(hit), w, h, threads, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"rmo_shadowmap_directional failed: {rc}")
    return {"depth": depth, "hit": hit, "stats": {"n_hit": st.n_hit, "sum_steps_p": st.sum_steps_p}}
