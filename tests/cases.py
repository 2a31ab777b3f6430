"""Named parity cases: each builds the material bindings (through the package's own
host-side scripts, i.e. the path a user takes) plus the inputs, so that the golden
generator, the CPU tests and the GPU tests all see identical uniforms."""
from __future__ import annotations

import math

import numpy as np

from raymarching_amd import Material, Raymarching, ShadowMap, ShadowMapDirectional, scenes
from raymarching_amd.unity import Camera, F, Transform, quat_angle_axis, quat_mul, vec3


def _base(mat: Material, w: int, h: int, camera=None, transforms=None) -> Raymarching:
    rm = Raymarching(camera or scenes.default_camera(w, h), mat, transforms or scenes.default_transforms())
    rm.bind()
    mat.SetVector("_WorldSpaceLightPos0", scenes.WORLD_SPACE_LIGHT_POS0)
    return rm


def transformed_transforms(pow2_scales: bool = False, unsafe_scale: bool = False):
    """the four slots with rotations and non-unit scales (general rotate()/divide path)"""
    ts = scenes.default_transforms()
    rots = [quat_angle_axis(33.0, (1, 2, 3)), quat_angle_axis(-71.0, (0, 1, 1)),
            quat_mul(quat_angle_axis(20.0, (0, 0, 1)), quat_angle_axis(12.0, (1, 0, 0))), quat_angle_axis(110.0, (1, 0, 0))]
    scl = [(1.3, 0.8, 1.1), (0.7, 1.9, 1.2), (0.9, 1.1, 1.05), (1.6, 0.6, 1.0)]
    if pow2_scales:
        scl = [(2.0, 1.0, 0.5), (0.5, 2.0, 1.0), (1.0, 1.0, 1.0), (2.0, 0.5, 1.0)]
    if unsafe_scale:
        # a significand of all ones is excluded from the reciprocal-based divide: true division path
        scl = [(1.3, float(np.float32(1.9999999)), 1.1), (0.7, 1.9, 1.2), (0.9, 1.1, 1.05), (1.6, 0.6, 1.0)]
    for t, q, s in zip(ts, rots, scl):
        t.rotation = q.astype(F)
        t.lossyScale = np.asarray(s, dtype=F)
    return ts


def synthetic_src(w: int, h: int) -> np.ndarray:
    y, x = np.mgrid[0:h, 0:w]
    img = np.empty((h, w, 4), dtype=np.uint8)
    img[..., 0] = (x * 7 + y * 3) & 255
    img[..., 1] = (x ^ y) & 255
    img[..., 2] = (x * y) & 255
    img[..., 3] = (x + y) & 255
    return img


def synthetic_depth(w: int, h: int) -> np.ndarray:
    """a mesh-depth texture: a wall at eye depth 11.3 over the left third, far elsewhere"""
    d = np.full((h, w), 1000.0, dtype=np.float32)
    d[:, : w // 3] = 11.3
    d[h // 2:, w // 3: w // 2] = 10.9
    return d


light_camera, light_rectangle = scenes.light_camera, scenes.light_rectangle


class Case:
    def __init__(self, name, w, h, kind="frag", setup=None, prims=None, src=False, sdepth=False, heavy=False, shadow=None):
        self.name, self.w, self.h, self.kind = name, w, h, kind
        self._setup, self.prims, self.use_src, self.use_sdepth, self.heavy = setup, prims, src, sdepth, heavy
        self.corners = None
        self.forward = None
        # frag cases that read a shadow map instead of marching the shadow ray: ("persp" | "dir", map width, map height)
        self.shadow = shadow

    def bind_shadow_pass(self, mat: Material):
        """binds `mat` (already bound for the frame) for the depth pass that feeds this case's shadow map; returns what the
        pass needs beyond the material: (corners, forward) for the directional pass, None for the perspective one"""
        kind, sw, sh = self.shadow
        if kind == "persp":
            sm = ShadowMap(light_camera(sw, sh), mat)
            mat.SetMatrix("_FrustrumCorners", sm.GetFrustumCorners())
            mat.SetMatrix("_CameraInvViewMatrix", sm.cam.cameraToWorldMatrix)
            mat.SetVector("_WorldSpaceCameraPos", sm.cam.transform.position)
            return None
        smd = ShadowMapDirectional(light_rectangle(), mat)
        smd.Update()
        return np.stack(smd.corners).astype(F).reshape(12), smd.transform.forward.astype(F)

    def bind_shadow_lookup(self, mat: Material) -> None:
        """what Raymarching.cs:92-93 would bind next to the texture itself"""
        kind, sw, sh = self.shadow
        if kind == "persp":
            cam = light_camera(sw, sh)
            mat.SetMatrix("_ShadowMapVPMatrix", cam.projectionMatrix * cam.worldToCameraMatrix)
            mat.SetFloat("_RmShadowMode", 1.0)
        else:
            smd = ShadowMapDirectional(light_rectangle(), mat)
            smd.Update()
            from raymarching_amd.unity import Matrix4x4
            k = Matrix4x4.identity()
            for i in range(4):
                k.SetRow(i, smd.corners[i])
            mat.SetMatrix("_RmShadowMapCorners", k)
            mat.SetVector("_RmShadowMapForward", smd.transform.forward)
            mat.SetFloat("_RmShadowMode", 2.0)

    def bind(self, mat: Material):
        """applies the case to a material; returns (src, scene_depth)"""
        if self._setup:
            self._setup(self, mat)
        else:
            _base(mat, self.w, self.h)
        if self.prims is not None:
            mat.SetPrimitives(self.prims)
        src = synthetic_src(self.w, self.h) if self.use_src else None
        sd = synthetic_depth(self.w, self.h) if self.use_sdepth else None
        return src, sd


def _orbit(k):
    def f(case, mat):
        _base(mat, case.w, case.h, camera=scenes.orbit_camera(k, 64, case.w, case.h))
    return f


def _transformed(pow2, unsafe=False):
    def f(case, mat):
        _base(mat, case.w, case.h, transforms=transformed_transforms(pow2, unsafe))
    return f


def _stress(case, mat):
    _base(mat, case.w, case.h, camera=scenes.stress_camera(case.w, case.h))
    for k, v in scenes.STRESS_FLOATS.items():
        mat.SetFloat(k, v)


def _fractional_steps(case, mat):
    _base(mat, case.w, case.h)
    mat.SetFloat("_MaxSteps", 37.5)          # slider values may be fractional: iterate while (float)i < steps
    mat.SetFloat("_MaxDist", 30.0)
    mat.SetFloat("_StepFactor", 0.35)
    mat.SetFloat("_NormalSampleScale", 0.001)


def _fractal(body, closeup, scale=None, rot=None, fscale=None, power=None):
    """a fractal scene() body around slot 0; closeup: camera 3.4 units from it instead of the default view"""
    def f(case, mat):
        ts = scenes.default_transforms()
        if scale is not None:
            ts[0].lossyScale = np.asarray(scale, dtype=F)
        if rot is not None:
            ts[0].rotation = rot.astype(F)
        cam = None
        if closeup:
            cam = scenes.closeup_camera(ts[0].position, case.w, case.h)
        _base(mat, case.w, case.h, camera=cam, transforms=ts)
        mat.SetFloat("_RmSceneBody", body)
        if fscale is not None:
            mat.SetFloat("_FractalScale", fscale)
        if power is not None:
            mat.SetFloat("_RmFractalPower", power)
    return f


def _composite(t, animate=False, floats=None, near=1.0):
    """_RmSceneBody 6, the line the reference keeps commented out at cginc:349, at _Time.y = t: a camera 3.4 units from the
    origin, where the bulb sits (the body reads the world-space p, no wrapper)"""
    def f(case, mat):
        cam = scenes.closeup_camera((0.0, 0.0, 0.0), case.w, case.h)
        cam.transform.position = (cam.transform.position.astype(np.float64) * near).astype(F)      # same direction, nearer
        _base(mat, case.w, case.h, camera=cam)
        mat.SetFloat("_RmSceneBody", 6.0)
        mat.SetVector("_Time", (t / 20.0, t, 2.0 * t, 3.0 * t))
        mat.SetFloat("_RmAnimateFractal", 1.0 if animate else 0.0)
        for k, v in (floats or {}).items():
            mat.SetFloat(k, v)
    return f


def _terrain(scale=1e-4, y0=-3.0, rot=None, floats=None):
    """the terrain body (cginc:315-330 as a height field) under the default camera: slot 0 scaled so that the 30000-unit
    heights and the 1.2e-5 noise frequency of the reference's formula land in the scene's range"""
    def f(case, mat):
        ts = scenes.default_transforms()
        ts[0].position = np.asarray([0.0, y0, 0.0], dtype=F)
        ts[0].lossyScale = np.asarray([scale, scale, scale], dtype=F)
        if rot is not None:
            ts[0].rotation = rot.astype(F)
        _base(mat, case.w, case.h, transforms=ts)
        mat.SetFloat("_RmSceneBody", 5.0)
        for k, v in (floats or {}).items():
            mat.SetFloat(k, v)
    return f


def _lit(floats, light_position=(-2.0, 6.0, -4.0), inner=None):
    """helpers the reference defines and never calls (getbrightness*, getscatter, refract2) switched on, with the
    `light l` of frag (shader:157-162) at _LightPosition"""
    def f(case, mat):
        if inner:
            inner(case, mat)
        else:
            _base(mat, case.w, case.h)
        mat.SetVector("_LightPosition", np.asarray(list(light_position) + [0.0], dtype=F))
        for k, v in floats.items():
            mat.SetFloat(k, v)
    return f


def _edge(floats=None, vectors=None):
    """default scene with degenerate uniforms"""
    def f(case, mat):
        _base(mat, case.w, case.h)
        for k, v in (floats or {}).items():
            mat.SetFloat(k, v)
        for k, v in (vectors or {}).items():
            mat.SetVector(k, np.asarray(v, dtype=F))
    return f


def _fx(floats, inner=None, bg=None):
    """the default scene (or another setup) with some of frag's commented-out lines switched on"""
    def f(case, mat):
        if inner:
            inner(case, mat)
        else:
            _base(mat, case.w, case.h)
        for k, v in floats.items():
            mat.SetFloat(k, v)
        if bg is not None:
            mat.SetVector("_BackgroundColor", np.asarray(bg, dtype=F))
    return f


def _domain(op, period=(0, 0, 0), size=1.0, inner=None, floats=None):
    """a scene seen through one of the domain operators (cginc:84-99, 156-165)"""
    def f(case, mat):
        if inner:
            inner(case, mat)
        else:
            _base(mat, case.w, case.h)
        mat.SetFloat("_RmDomainOp", float(op))
        mat.SetVector("_RmDomainPeriod", np.asarray(list(period) + [0.0], dtype=F))
        mat.SetFloat("_RmDomainSize", float(size))
        for k, v in (floats or {}).items():
            mat.SetFloat(k, v)
    return f


def _shadow_persp(case, mat):
    _base(mat, case.w, case.h)
    sm = ShadowMap(scenes.shadow_camera(case.w, case.h), mat)
    m = sm.shadowMapMaterial
    m.SetMatrix("_FrustrumCorners", sm.GetFrustumCorners())
    m.SetMatrix("_CameraInvViewMatrix", sm.cam.cameraToWorldMatrix)
    m.SetVector("_WorldSpaceCameraPos", sm.cam.transform.position)


def _shadow_dir(case, mat):
    _base(mat, case.w, case.h)
    # the ShadowMap object's transform reused as the rectangle (the 4K depth-pass benchmark configuration):
    # a 14 x 8 rectangle above and behind the objects, looking back at them
    t = Transform(position=vec3(-3, 3, 9), rotation=scenes.shadow_camera().transform.rotation, lossyScale=vec3(14, 8, 1))
    smd = ShadowMapDirectional(t, mat)
    smd.Update()
    case.corners = np.stack(smd.corners).astype(F).reshape(12)
    case.forward = t.forward.astype(F)


CASES = [
    Case("default_640x360", 640, 360),
    Case("default_1920x1080", 1920, 1080, heavy=True),
    Case("default_3840x2160", 3840, 2160, heavy=True),
    Case("default_src_depth_333x187", 333, 187, src=True, sdepth=True),      # ragged size, _MainTex + depth texture
    Case("default_src_depth_640x360", 640, 360, src=True, sdepth=True),      # what Graphics.Blit(src, dest, mat) passes, at a size with exact reciprocals
    Case("orbit_k5_480x270", 480, 270, setup=_orbit(5)),
    Case("orbit_k23_480x270", 480, 270, setup=_orbit(23)),
    Case("orbit_k40_480x270", 480, 270, setup=_orbit(40)),
    Case("transformed_400x225", 400, 225, setup=_transformed(False)),
    Case("transformed_pow2_400x225", 400, 225, setup=_transformed(True)),
    Case("transformed_truediv_400x225", 400, 225, setup=_transformed(False, True)),
    Case("fractional_steps_320x180", 320, 180, setup=_fractional_steps),
    Case("tiny_1x1", 1, 1),
    Case("tiny_17x3", 17, 3),
    Case("stress256_320x180", 320, 180, setup=_stress, prims=scenes.stress_primitives(256)),
    Case("stress13_200x112", 200, 112, setup=_stress, prims=scenes.stress_primitives(13, seed=0x5EED000D)),
    # tables the cooperative kernel splits over four waves: 33 -> chunks of 16, 16, 1 (three chunks); 100 -> 32, 32, 32, 4;
    # 64 untransformed primitives (the variant without rotation and scale); a depth pass over a table
    Case("stress33_160x90", 160, 90, setup=_stress, prims=scenes.stress_primitives(33, seed=0x5EED0021)),
    Case("stress100_160x90", 160, 90, setup=_stress, prims=scenes.stress_primitives(100, seed=0x5EED0064)),
    Case("stress64_untransformed_160x90", 160, 90, setup=_stress,
         prims=[dict(p, quat=(0.0, 0.0, 0.0, 1.0), scale=(1.0, 1.0, 1.0)) for p in scenes.stress_primitives(64, seed=0x5EED0040)]),
    Case("stress48_shadow_dir_128x72", 128, 72, kind="shadow_dir", setup=_shadow_dir, prims=scenes.stress_primitives(48, seed=0x5EED0030)),
    Case("prims4_as_table_320x180", 320, 180, prims=[
        {"type": 0, "pos": (-0.584, 0.049, 0.602), "quat": (0, 0, 0, 1), "scale": (1, 1, 1), "color": (1, .25, .25), "params": (1,)},
        {"type": 1, "pos": (-0.74, 0.929, 0.0), "quat": (0, 0, 0, 1), "scale": (1, 1, 1), "color": (.25, 1, .25), "params": (1, 1, 1)},
        {"type": 2, "pos": (-7.87, 0.0, 1.522), "quat": (0, 0, 0, 1), "scale": (1, 1, 1), "color": (.25, .25, 1), "params": (5, 5, .1)},
        {"type": 3, "pos": (-3.074, -0.087, 0.08), "quat": (0, 0, 0, 1), "scale": (1, 1, 1), "color": (1, 1, .25), "params": (.5, .5)},
    ]),
    Case("csg7_480x270", 480, 270, prims=scenes.csg_primitives()),
    Case("csg_random37_240x135", 240, 135, setup=_stress, prims=scenes.csg_random_primitives(37)),
    Case("csg_random70_160x90", 160, 90, setup=_stress, prims=scenes.csg_random_primitives(70, seed=0x5EED070)),
    # op bits in tables the cooperative kernel splits: 64 = four full chunks, 100 = 32, 32, 32, 4 (37 and 70 above: three chunks)
    # the smallest tables that split over four waves: 3 = one primitive per wave and an idle one, 5 = chunks of 2, 2, 1
    Case("csg_random3_96x54", 96, 54, setup=_stress, prims=scenes.csg_random_primitives(3, seed=0x5EED003)),
    Case("csg_random5_96x54", 96, 54, setup=_stress, prims=scenes.csg_random_primitives(5, seed=0x5EED005)),
    Case("csg_random64_128x72", 128, 72, setup=_stress, prims=scenes.csg_random_primitives(64, seed=0x5EED064)),
    Case("csg_random100_128x72", 128, 72, setup=_stress, prims=scenes.csg_random_primitives(100, seed=0x5EED100)),
    Case("menger_closeup_320x180", 320, 180, setup=_fractal(1, True)),
    Case("menger_scaled_rotated_256x144", 256, 144, setup=_fractal(1, False, scale=(3.0, 2.5, 3.5), rot=quat_angle_axis(33.0, (1, 2, 3)))),
    Case("sierpinski_closeup_320x180", 320, 180, setup=_fractal(2, True)),
    Case("sierpinski_scale1p7_240x135", 240, 135, setup=_fractal(2, False, scale=(2.0, 2.0, 2.0), fscale=1.7)),
    # degenerate uniforms: the loops must stop (or not start) exactly where the shader's would
    Case("edge_zero_steps_64x36", 64, 36, setup=_edge({"_MaxSteps": 0.0})),
    Case("edge_negative_steps_64x36", 64, 36, setup=_edge({"_MaxSteps": -3.0})),
    Case("edge_one_step_64x36", 64, 36, setup=_edge({"_MaxSteps": 1.0})),
    Case("edge_zero_dist_64x36", 64, 36, setup=_edge({"_MaxDist": 0.0})),
    Case("edge_zero_stepfactor_64x36", 64, 36, setup=_edge({"_StepFactor": 0.0, "_MaxSteps": 50.0})),
    Case("edge_negative_stepfactor_64x36", 64, 36, setup=_edge({"_StepFactor": -0.7, "_MaxSteps": 50.0})),
    Case("edge_zero_normal_scale_128x72", 128, 72, setup=_edge({"_NormalSampleScale": 0.0})),
    Case("edge_negative_normal_scale_128x72", 128, 72, setup=_edge({"_NormalSampleScale": -1.0})),
    Case("edge_zero_light_128x72", 128, 72, setup=_edge(vectors={"_WorldSpaceLightPos0": (0, 0, 0, 0)})),
    Case("edge_light_from_below_128x72", 128, 72, setup=_edge(vectors={"_WorldSpaceLightPos0": (0.3, -0.8, 0.2, 0)})),
    Case("edge_camera_inside_box_128x72", 128, 72, setup=_edge(vectors={"_WorldSpaceCameraPos": (-0.74, 0.929, 0.0, 0)})),
    Case("edge_big_steps_2000_cap_96x54", 96, 54, setup=_edge({"_MaxSteps": 5000.0, "_StepFactor": 0.01, "_MaxDist": 30.0})),
    Case("edge_nan_camera_32x18", 32, 18, setup=_edge({"_MaxSteps": 20.0}, vectors={"_WorldSpaceCameraPos": (float("nan"), 1, -11, 0)})),
    Case("edge_inf_camera_32x18", 32, 18, setup=_edge({"_MaxSteps": 20.0}, vectors={"_WorldSpaceCameraPos": (float("inf"), 1, -11, 0)})),
    Case("tall_3x257", 3, 257),
    Case("wide_1025x2", 1025, 2),
    Case("mandelbulb_closeup_240x135", 240, 135, setup=_fractal(3, True)),
    Case("mandelbulb_power8_rotated_200x112", 200, 112, setup=_fractal(3, False, scale=(3.0, 3.0, 3.0), rot=quat_angle_axis(-50.0, (1, 0.3, 0.2)), power=8.0)),
    Case("mandelbrot_closeup_240x135", 240, 135, setup=_fractal(4, True)),
    Case("mandelbrot_power3_200x112", 200, 112, setup=_fractal(4, False, scale=(2.5, 2.5, 2.5), power=3.0)),
    # frag's commented-out lines (shader:178-187, 198-203, 206-212) switched on
    Case("fx_step_heat_320x180", 320, 180, setup=_fx({"_RmStepHeat": 1.0})),
    Case("fx_view_shadow_160x90", 160, 90, setup=_fx({"_RmDebugView": 1.0})),
    Case("fx_view_normal_160x90", 160, 90, setup=_fx({"_RmDebugView": 2.0})),
    Case("fx_view_red_160x90", 160, 90, setup=_fx({"_RmDebugView": 3.0})),
    # the integrated light is ~2.5 in free space: a background of -2.4 keeps the sum inside [0,1] where it shows
    Case("fx_volumetric_96x54", 96, 54, setup=_fx({"_RmVolumetricLight": 1.0}, bg=(-2.4, -2.3, -2.2, 1.0))),
    Case("fx_bounce_320x180", 320, 180, setup=_fx({"_RmBounce": 1.0})),
    Case("fx_all_transformed_128x72", 128, 72, setup=_fx({"_RmBounce": 1.0, "_RmStepHeat": 1.0, "_RmVolumetricLight": 1.0},
                                                         inner=_transformed(False), bg=(-2.6, -2.5, -2.4, 1.0))),
    Case("fx_heat_bounce_stress13_160x90", 160, 90, setup=_fx({"_RmBounce": 1.0, "_RmStepHeat": 1.0}, inner=_stress),
         prims=scenes.stress_primitives(13, seed=0x5EED000D)),
    Case("fx_bounce_view_menger_160x90", 160, 90, setup=_fx({"_RmBounce": 1.0, "_RmDebugView": 2.0}, inner=_fractal(1, True))),
    # the shadow map read in the frame pass instead of the soft-shadow march (what _ShadowMap/_ShadowMapVPMatrix are for)
    Case("shadowmap_lookup_persp_320x180", 320, 180, shadow=("persp", 512, 512)),
    Case("shadowmap_lookup_dir_320x180", 320, 180, shadow=("dir", 640, 384)),
    Case("shadowmap_lookup_dir_transformed_bounce_200x112", 200, 112, shadow=("dir", 300, 200),
         setup=_fx({"_RmBounce": 1.0, "_RmShadowBias": 0.03}, inner=_transformed(False))),
    Case("shadowmap_lookup_persp_stress13_lowres_160x90", 160, 90, shadow=("persp", 64, 48), setup=_stress,
         prims=scenes.stress_primitives(13, seed=0x5EED000D)),
    # terrain / noise / rand (cginc:125-149, 315-330) as a scene() body
    Case("composite_t0_200x112", 200, 112, setup=_composite(0.0)),
    Case("composite_t0p8_240x135", 240, 135, setup=_composite(0.8)),
    Case("composite_near_t1p7_200x112", 200, 112, setup=_composite(1.7, near=0.42)),
    Case("composite_animated_t2p3_bounce_160x90", 160, 90, setup=_composite(2.3, True, {"_RmBounce": 1.0, "_RmStepHeat": 1.0})),
    Case("menger_animated_t1p1_200x112", 200, 112, setup=lambda c, m: (_fractal(1, True)(c, m), m.SetVector("_Time", (0.055, 1.1, 2.2, 3.3)),
                                                                       m.SetFloat("_RmAnimateFractal", 1.0))),
    Case("terrain_320x180", 320, 180, setup=_terrain()),
    Case("terrain_rotated_nonuniform_200x112", 200, 112, setup=_terrain(scale=2e-4, y0=-4.0, rot=quat_angle_axis(12.0, (1, 0, 0.3)),
                                                                        floats={"_StepFactor": 0.4})),
    Case("terrain_shadow_dir_128x72", 128, 72, kind="shadow_dir", setup=lambda c, m: (_shadow_dir(c, m), _terrain()(c, m))),
    # getbrightness / getbrightnesshard / getscatter / refract2 (shader:84-100, 129-147) wired in
    Case("fx_getbrightness_240x135", 240, 135, setup=_lit({"_RmLightModel": 1.0})),
    Case("fx_getbrightnesshard_240x135", 240, 135, setup=_lit({"_RmLightModel": 2.0}, light_position=(-3.0, 3.0, -2.5))),
    Case("fx_getscatter_240x135", 240, 135, setup=_lit({"_RmScatter": 1.0}, light_position=(-3.0, 2.0, -3.0))),
    Case("fx_refract2_bounce_scatter_transformed_200x112", 200, 112,
         setup=_lit({"_RmBounce": 2.0, "_RefractiveIndex": 1.3, "_RmScatter": 1.0, "_RmLightModel": 1.0}, inner=_transformed(False))),
    Case("shadow_persp_100x100", 100, 100, kind="shadow_persp", setup=_shadow_persp),
    Case("shadow_dir_256x144", 256, 144, kind="shadow_dir", setup=_shadow_dir),
    # domain operators in front of scene()
    Case("dom_repeat_xz_320x180", 320, 180, setup=_domain(1, period=(9.0, 0.0, 7.5))),
    Case("dom_repeat_xyz_stress13_160x90", 160, 90, setup=_domain(1, period=(8.0, 40.0, 8.0), inner=_stress),
         prims=scenes.stress_primitives(13, seed=0x5EED000D)),
    Case("dom_repeat_menger_bounce_160x90", 160, 90, setup=_domain(1, period=(3.0, 3.0, 3.0), inner=_fractal(1, True),
                                                                    floats={"_RmBounce": 1.0})),
    Case("dom_boxfold_240x135", 240, 135, setup=_domain(2, size=6.0)),
    Case("dom_boxfold_sierpinski_160x90", 160, 90, setup=_domain(2, size=2.5, inner=_fractal(2, True))),
    Case("dom_spherefold_240x135", 240, 135, setup=_domain(3, size=3.0)),
    Case("dom_repeat_shadow_dir_128x72", 128, 72, kind="shadow_dir", setup=lambda c, m: (_shadow_dir(c, m), _domain(1, period=(6.0, 0.0, 6.0), inner=lambda c2, m2: None)(c, m))),
    # the benchmark configurations at their full sizes (hashes only; checked on the GPU)
    Case("shadow_dir_3840x2160", 3840, 2160, kind="shadow_dir", setup=_shadow_dir, heavy=True),
    Case("orbit_k17_7680x4320", 7680, 4320, setup=_orbit(17), heavy=True),
    Case("stress256_1920x1080", 1920, 1080, setup=_stress, prims=scenes.stress_primitives(256), heavy=True),
]

BY_NAME = {c.name: c for c in CASES}


def _stress_cam(position, euler_deg=(2.0, 0.0, 0.0), floats=None, vectors=None):
    """the stress-table set-up seen from another camera, optionally with other march uniforms"""
    def f(case, mat):
        rot = quat_mul(quat_angle_axis(euler_deg[1], (0, 1, 0)), quat_angle_axis(euler_deg[0], (1, 0, 0)))
        cam = Camera(transform=Transform(position=vec3(*position), rotation=rot), fieldOfView=80.0, aspect=case.w / case.h,
                     nearClipPlane=0.3, farClipPlane=100000.0)
        _base(mat, case.w, case.h, camera=cam)
        for k, v in dict(scenes.STRESS_FLOATS, **(floats or {})).items():
            mat.SetFloat(k, v)
        for k, v in (vectors or {}).items():
            mat.SetVector(k, np.asarray(v, dtype=F))
    return f


def _edited(prims, edits):
    out = [dict(p) for p in prims]
    for i, e in edits.items():
        out[i] = dict(out[i], **e)
    return out


# Plain smooth-union tables of 33 to 512 primitives take the shared-tile kernel with exact culling in the marches: table
# sizes that give full, partial and one-leaf quarters, the variants without rotation / scale, a camera inside the slab, the
# guards (coordinates beyond the range the rounding margin is sized for, a negative and a zero scale, an enormous
# primitive, a camera that is not a number) and a depth pass.  Compared with the oracle on the GPU only (no golden entry).
SKIP_CASES = [
    Case("skip_stress65_160x90", 160, 90, setup=_stress, prims=scenes.stress_primitives(65, seed=0x5EED0041)),
    Case("skip_stress128_160x90", 160, 90, setup=_stress, prims=scenes.stress_primitives(128, seed=0x5EED0080)),
    Case("skip_stress200_160x90", 160, 90, setup=_stress, prims=scenes.stress_primitives(200, seed=0x5EED00C8)),
    # quarters of 128 leaves: two leaves per lane in the bounds, two mask words (257: 128, 128 and a single leaf)
    Case("skip_stress257_128x72", 128, 72, setup=_stress, prims=scenes.stress_primitives(257, seed=0x5EED0101)),
    Case("skip_stress300_128x72", 128, 72, setup=_stress, prims=scenes.stress_primitives(300, seed=0x5EED012C)),
    Case("skip_stress512_128x72", 128, 72, setup=_stress, prims=scenes.stress_primitives(512, seed=0x5EED0200)),
    Case("skip_stress512_inside_96x54", 96, 54, setup=_stress_cam((0.3, 0.1, -2.0), (0.0, 25.0, 0.0)), prims=scenes.stress_primitives(512, seed=0x5EED0200)),
    Case("skip_stress256_inside_160x90", 160, 90, setup=_stress_cam((0.3, 0.1, -2.0), (0.0, 25.0, 0.0)), prims=scenes.stress_primitives(256)),
    Case("skip_stress256_above_128x72", 128, 72, setup=_stress_cam((0.0, 9.0, 0.0), (88.0, 0.0, 0.0)), prims=scenes.stress_primitives(256)),
    Case("skip_stress128_far_camera_96x54", 96, 54, prims=scenes.stress_primitives(128, seed=0x5EED0080),
         setup=_stress_cam((0.0, 1100.0, 0.0), (90.0, 0.0, 0.0), floats={"_MaxDist": 1200.0, "_MaxSteps": 64.0, "_StepFactor": 0.9})),
    Case("skip_stress128_untransformed_128x72", 128, 72, setup=_stress,
         prims=[dict(p, quat=(0.0, 0.0, 0.0, 1.0), scale=(1.0, 1.0, 1.0)) for p in scenes.stress_primitives(128, seed=0x5EED0080)]),
    Case("skip_stress128_scaled_only_128x72", 128, 72, setup=_stress,
         prims=[dict(p, quat=(0.0, 0.0, 0.0, 1.0)) for p in scenes.stress_primitives(128, seed=0x5EED0080)]),
    Case("skip_stress128_rotated_only_128x72", 128, 72, setup=_stress,
         prims=[dict(p, scale=(1.0, 1.0, 1.0)) for p in scenes.stress_primitives(128, seed=0x5EED0080)]),
    Case("skip_stress128_negative_scale_128x72", 128, 72, setup=_stress,
         prims=_edited(scenes.stress_primitives(128, seed=0x5EED0080), {5: {"scale": (-1.2, 0.8, 1.0)}, 77: {"scale": (0.7, -0.6, -1.1)}})),
    Case("skip_stress128_zero_scale_96x54", 96, 54, setup=_stress,
         prims=_edited(scenes.stress_primitives(128, seed=0x5EED0080), {40: {"scale": (1.0, 0.0, 1.0)}})),
    Case("skip_stress128_enormous_sphere_96x54", 96, 54, setup=_stress,
         prims=_edited(scenes.stress_primitives(128, seed=0x5EED0080), {64: {"type": 0, "pos": (0.0, -40000.0, 0.0), "scale": (1.0, 1.0, 1.0),
                                                                            "params": (79990.0, 0, 0, 0)}})),
    Case("skip_stress128_nan_camera_32x18", 32, 18, prims=scenes.stress_primitives(128, seed=0x5EED0080),
         setup=_stress_cam((float("nan"), 0.75, -14.0), floats={"_MaxSteps": 12.0})),
    Case("skip_stress128_shadow_dir_128x72", 128, 72, kind="shadow_dir", setup=_shadow_dir, prims=scenes.stress_primitives(128, seed=0x5EED0080)),
]
BY_NAME.update({c.name: c for c in SKIP_CASES})


def _scaled_unrotated(scl, floats=None, vectors=None, quats=None):
    """the four slots with non-unit scales and no rotation: the kernel variants whose soft-shadow march skips the (sphere, box)
    pair while it provably cannot change a bit of the step's result"""
    def f(case, mat):
        ts = scenes.default_transforms()
        for i, (t, s_) in enumerate(zip(ts, scl)):
            t.lossyScale = np.asarray(s_, dtype=F)
            if quats is not None:
                t.rotation = np.asarray(quats[i], dtype=F)
        _base(mat, case.w, case.h, transforms=ts)
        for k, v in (floats or {}).items():
            mat.SetFloat(k, v)
        for k, v in (vectors or {}).items():
            mat.SetVector(k, np.asarray(v, dtype=F))
    return f


def _offset_scene(offset, scl=None):
    """the 4-slot scene and its camera moved away from the origin: sample points with large coordinates (coarse ulps)"""
    def f(case, mat):
        off = np.asarray(offset, dtype=F)
        ts = scenes.default_transforms()
        for i, t in enumerate(ts):
            t.position = (np.asarray(t.position, dtype=F) + off).astype(F)
            if scl is not None:
                t.lossyScale = np.asarray(scl[i], dtype=F)
        cam = scenes.default_camera(case.w, case.h)
        cam.transform.position = (np.asarray(cam.transform.position, dtype=F) + off).astype(F)
        _base(mat, case.w, case.h, camera=cam, transforms=ts)
    return f


_ANISO = [(1.3, 0.8, 1.1), (0.7, 1.9, 1.2), (0.9, 1.1, 1.05), (1.6, 0.6, 1.0)]
# The pair skip of the 4-slot soft-shadow march: scales of the reciprocal, power-of-two and true-division variants, lights that
# make long shadow marches, and the guards (a negative scale, a zero scale, an infinite scale, a quaternion that is not of unit
# length: the skip must then stay off and the frame still be the oracle's).  Compared with the oracle on the GPU only.
PAIR_SKIP_CASES = [
    Case("pair_skip_scaled_400x225", 400, 225, setup=_scaled_unrotated(_ANISO)),
    Case("pair_skip_pow2_320x180", 320, 180, setup=_scaled_unrotated([(2.0, 1.0, 0.5), (0.5, 2.0, 1.0), (1.0, 1.0, 1.0), (2.0, 0.5, 1.0)])),
    Case("pair_skip_truediv_320x180", 320, 180,
         setup=_scaled_unrotated([(1.3, float(np.float32(1.9999999)), 1.1), (0.7, 1.9, 1.2), (0.9, 1.1, 1.05), (1.6, 0.6, 1.0)])),
    Case("pair_skip_low_light_320x180", 320, 180, setup=_scaled_unrotated(_ANISO, vectors={"_WorldSpaceLightPos0": (0.9, 0.12, -0.4, 0)})),
    Case("pair_skip_unit_low_light_480x270", 480, 270, setup=_edge(vectors={"_WorldSpaceLightPos0": (-0.8, 0.1, 0.55, 0)})),
    Case("pair_skip_long_light_vector_320x180", 320, 180, setup=_edge(vectors={"_WorldSpaceLightPos0": (1.4, 2.5, -1.9, 0)})),
    Case("pair_skip_small_steps_240x135", 240, 135, setup=_scaled_unrotated(_ANISO, floats={"_StepFactor": 0.2, "_MaxSteps": 900.0})),
    Case("pair_skip_big_scene_240x135", 240, 135, setup=_scaled_unrotated([(4.0, 4.0, 4.0), (3.0, 5.0, 2.0), (1.0, 1.0, 1.0), (6.0, 0.5, 6.0)])),
    Case("pair_skip_offset_600_320x180", 320, 180, setup=_offset_scene((600.0, -300.0, 450.0))),
    Case("pair_skip_offset_600_scaled_240x135", 240, 135, setup=_offset_scene((-610.0, 12.0, 580.0), _ANISO)),
    Case("pair_skip_offset_30000_240x135", 240, 135, setup=_offset_scene((30000.0, 0.0, -30000.0))),
    Case("pair_skip_offset_4e6_160x90", 160, 90, setup=_offset_scene((4.0e6, 1.0e6, 4.0e6))),
    Case("pair_skip_negative_scale_200x112", 200, 112, setup=_scaled_unrotated([(1.3, -0.8, 1.1), (0.7, 1.9, 1.2), (0.9, 1.1, 1.05), (1.6, 0.6, 1.0)])),
    Case("pair_skip_zero_scale_200x112", 200, 112, setup=_scaled_unrotated([(1.3, 0.8, 1.1), (0.7, 0.0, 1.2), (0.9, 1.1, 1.05), (1.6, 0.6, 1.0)])),
    Case("pair_skip_inf_scale_200x112", 200, 112, setup=_scaled_unrotated([(float("inf"), 0.8, 1.1), (0.7, 1.9, 1.2), (0.9, 1.1, 1.05), (1.6, 0.6, 1.0)])),
    Case("pair_skip_long_quaternion_200x112", 200, 112,
         setup=_scaled_unrotated(_ANISO, quats=[(0.0, 0.0, 0.0, 1.5), (0.0, 0.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)])),
]
BY_NAME.update({c.name: c for c in PAIR_SKIP_CASES})


def oracle_run(case: Case, O, threads: int = 0) -> dict:
    """binds the case on a fresh material and renders it with the CPU oracle"""
    with Material() as mat:
        src, sd = case.bind(mat)
        shadow_map = None
        if case.shadow:
            with Material() as sm_mat:                      # the depth pass sees the same scene through the shadow camera
                case.bind(sm_mat)
                extra = case.bind_shadow_pass(sm_mat)
                su = O.Uniforms(O.props_from_material(sm_mat), prims=case.prims)
                kind, sw, sh = case.shadow
                shadow_map = (O.shadowmap_perspective(su, sw, sh, threads=threads) if kind == "persp"
                              else O.shadowmap_directional(su, extra[0], extra[1], sw, sh, threads=threads))["depth"]
            case.bind_shadow_lookup(mat)
        uni = O.Uniforms(O.props_from_material(mat), prims=case.prims, shadow_map=shadow_map)
        if case.kind == "frag":
            return O.render(uni, case.w, case.h, src=src, scene_depth=sd, threads=threads, want_steps=True)
        if case.kind == "shadow_persp":
            return O.shadowmap_perspective(uni, case.w, case.h, threads=threads)
        return O.shadowmap_directional(uni, case.corners, case.forward, case.w, case.h, threads=threads)


def random_case(seed: int) -> Case:
    """A seeded random scene for fuzzing: 1..40 primitives of all five kinds with random placement,
    rotation, scale and op bits, or the 4-slot scene with random transforms, or a fractal body; random
    camera on a sphere around the origin, random march uniforms, sometimes a source image / depth texture /
    optional frag lines / a domain operator."""
    rng = scenes.SplitMix64(0xF0220000 + seed)
    u = rng.uniform
    w, h = 24 + int(u() * 72), 16 + int(u() * 48)

    def quat():
        a = [u() * 2 - 1 for _ in range(4)]
        n = math.sqrt(sum(x * x for x in a)) or 1.0
        return [x / n for x in a]

    kind = int(u() * 4)            # 0 table, 1 table, 2 four slots, 3 fractal
    prims = None
    if kind <= 1:
        n = 1 + int(u() * 40)
        prims = []
        for i in range(n):
            t = int(u() * 5)
            params = [(0.5 + u(),), (0.5 + u(), 0.5 + u(), 0.5 + u()), (0.3 + u(), 0.2 + 0.5 * u(), 0.05 + 0.1 * u()),
                      (0.3 + 0.5 * u(), 0.3 + 0.5 * u()), (0.4 + 0.4 * u(), 0.1 + 0.2 * u())][t]
            ident = u() < 0.3
            unit = u() < 0.3
            op = 0
            if i > 0 and u() < 0.15:
                op |= scenes.OP_SMAX
            if u() < 0.1:
                op |= scenes.OP_NEGATE
            prims.append({"type": t, "op": op, "pos": [float(F(u() * 8 - 4)), float(F(u() * 3 - 1.5)), float(F(u() * 8 - 4))],
                          "quat": [0.0, 0.0, 0.0, 1.0] if ident else [float(F(x)) for x in quat()],
                          "scale": [1.0, 1.0, 1.0] if unit else [float(F(0.5 + 1.5 * u())) for _ in range(3)],
                          "color": [float(F(0.25 + 0.75 * u())) for _ in range(3)], "params": params})
    floats = {"_MaxSteps": float(int(20 + u() * 200)), "_MaxDist": float(F(10 + u() * 40)), "_StepFactor": float(F(0.3 + u() * 0.7)),
              "_NormalSampleScale": float(F(10 ** (-5 + 3 * u())))}
    if kind == 3:
        floats["_RmSceneBody"] = float(1 + int(u() * 5))
        floats["_FractalScale"] = float(F(1.5 + u()))
        floats["_MaxSteps"] = float(int(20 + u() * 60))
    fx = u()
    if fx < 0.1:
        floats["_RmStepHeat"] = 1.0
    elif fx < 0.2:
        floats["_RmBounce"] = 1.0
    elif fx < 0.3:
        floats["_RmDebugView"] = float(1 + int(u() * 3))
    elif fx < 0.36:
        floats["_RmLightModel"] = 1.0 if fx < 0.33 else 2.0
    elif fx < 0.42:
        floats["_RmScatter"] = 1.0
    elif fx < 0.46:
        floats["_RmBounce"], floats["_RefractiveIndex"] = 2.0, 1.25
    az, el, dist = u() * 2 * math.pi, (u() - 0.5) * 1.2, 4 + u() * 8
    pos = np.array([dist * math.cos(el) * math.sin(az), dist * math.sin(el), -dist * math.cos(el) * math.cos(az)])
    fov = 40 + u() * 60
    slot_rot = [quat() for _ in range(4)]
    slot_scl = [[0.5 + 1.5 * u() for _ in range(3)] for _ in range(4)]
    transformed = u() < 0.6
    light = [u() * 2 - 1, u(), u() * 2 - 1, 0.0]
    use_src, use_sd = u() < 0.3, u() < 0.3
    dom, dom_op, dom_period, dom_size = u(), 0, [0.0, 0.0, 0.0], 1.0
    if dom < 0.12:
        dom_op, dom_period = 1, [float(F(3 + 6 * u())) if u() < 0.7 else 0.0 for _ in range(3)]
    elif dom < 0.18:
        dom_op, dom_size = 2, float(F(2 + 6 * u()))
    elif dom < 0.24:
        dom_op, dom_size = 3, float(F(1 + 3 * u()))

    def setup(case, mat):
        from raymarching_amd.unity import Camera, quat_look_rotation
        cam = Camera(transform=Transform(position=pos.astype(F), rotation=quat_look_rotation(-pos)), fieldOfView=fov,
                     aspect=case.w / case.h, nearClipPlane=0.3, farClipPlane=1000.0)
        ts = scenes.default_transforms()
        if kind >= 2:
            for t in ts:                       # bring the four slots around the origin, where the camera looks
                t.position = (t.position * F(0.4)).astype(F)
        if transformed:
            for t, q, sc in zip(ts, slot_rot, slot_scl):
                t.rotation = np.asarray(q, dtype=F)
                t.lossyScale = np.asarray(sc, dtype=F)
        _base(mat, case.w, case.h, camera=cam, transforms=ts)
        mat.SetVector("_WorldSpaceLightPos0", np.asarray(light, dtype=F))
        mat.SetVector("_LightPosition", np.asarray([2.5, 4.0, -3.0, 0.0], dtype=F))
        for k, v in floats.items():
            mat.SetFloat(k, v)
        if dom_op:
            mat.SetFloat("_RmDomainOp", float(dom_op))
            mat.SetVector("_RmDomainPeriod", np.asarray(dom_period + [0.0], dtype=F))
            mat.SetFloat("_RmDomainSize", dom_size)

    return Case(f"random_{seed}", w, h, setup=setup, prims=prims, src=use_src, sdepth=use_sd)


def random_table_case(seed: int, world_scale: float = 1.0, csg: bool = False) -> Case:
    """A seeded random plain smooth union of 33..512 primitives of all five kinds (the tables whose marches cull exactly):
    dense or sparse placement, any rotation, scales in [.3, 2.3] (some unit, some identity rotations, sometimes the whole
    table without rotation or scale), random camera looking at the origin from inside or outside the cloud, random march
    uniforms, sometimes a source image / depth texture.  world_scale multiplies every length of the scene but the blend
    radius of smin: small worlds blend everything (nothing can be skipped), large ones put coordinates near and beyond the
    range the culling's rounding margin is sized for.  csg: some nodes are smax, some leaves negated."""
    rng = scenes.SplitMix64(0x7AB1E000 + seed + (0x1000 if csg else 0))
    u = rng.uniform
    w, h = 24 + int(u() * 56), 16 + int(u() * 32)

    def quat():
        a = [u() * 2 - 1 for _ in range(4)]
        n = math.sqrt(sum(x * x for x in a)) or 1.0
        return [x / n for x in a]

    n = 33 + int(u() * 480)
    ext = [2.0 + u() * 18.0, 0.5 + u() * 6.0, 2.0 + u() * 18.0]
    no_rot, no_scale = u() < 0.15, u() < 0.15
    prims = []
    for i in range(n):
        t = int(u() * 5)
        params = [(0.5 + u(),), (0.5 + u(), 0.5 + u(), 0.5 + u()), (0.3 + u(), 0.2 + 0.5 * u(), 0.05 + 0.1 * u()),
                  (0.3 + 0.5 * u(), 0.3 + 0.5 * u()), (0.4 + 0.4 * u(), 0.1 + 0.2 * u())][t]
        ident, unit = no_rot or u() < 0.2, no_scale or u() < 0.2
        params = tuple(float(F(v * world_scale)) for v in params)
        op = 0
        if csg and i > 0 and u() < 0.15:
            op |= scenes.OP_SMAX
        if csg and u() < 0.1:
            op |= scenes.OP_NEGATE
        prims.append({"type": t, "op": op, "pos": [float(F((u() * 2 - 1) * e * world_scale)) for e in ext],
                      "quat": [0.0, 0.0, 0.0, 1.0] if ident else [float(F(x)) for x in quat()],
                      "scale": [1.0, 1.0, 1.0] if unit else [float(F(0.3 + 2.0 * u())) for _ in range(3)],
                      "color": [float(F(0.25 + 0.75 * u())) for _ in range(3)], "params": params})
    floats = {"_MaxSteps": float(int(20 + u() * 150)), "_MaxDist": float(F((15 + u() * 60) * max(1.0, world_scale))),
              "_StepFactor": float(F(0.3 + u() * 0.7)), "_NormalSampleScale": float(F(10 ** (-5 + 3 * u())))}
    az, el = u() * 2 * math.pi, (u() - 0.5) * 1.4
    dist = (0.2 + u() * 1.8) * max(ext) * world_scale
    pos = np.array([dist * math.cos(el) * math.sin(az), dist * math.sin(el), -dist * math.cos(el) * math.cos(az)])
    fov = 40 + u() * 60
    light = [u() * 2 - 1, u(), u() * 2 - 1, 0.0]
    use_src, use_sd = u() < 0.2, u() < 0.2

    def setup(case, mat):
        from raymarching_amd.unity import quat_look_rotation
        cam = Camera(transform=Transform(position=pos.astype(F), rotation=quat_look_rotation(-pos)), fieldOfView=fov,
                     aspect=case.w / case.h, nearClipPlane=0.3, farClipPlane=1000.0)
        _base(mat, case.w, case.h, camera=cam)
        mat.SetVector("_WorldSpaceLightPos0", np.asarray(light, dtype=F))
        for k, v in floats.items():
            mat.SetFloat(k, v)

    return Case(f"random_table_{seed}_x{world_scale:g}{'_csg' if csg else ''}", w, h, setup=setup, prims=prims, src=use_src, sdepth=use_sd)
