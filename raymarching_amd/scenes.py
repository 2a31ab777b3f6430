"""The numeric configuration of the reference's sample scene
(Assets/Scenes/SampleScene.unity, Assets/Raymarching.mat) and the synthetic
workloads built on it (orbit poses, 256-primitive stress scene).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

from .unity import Camera, F, Transform, quat_angle_axis, quat_look_rotation, quat_mul, vec3

# material floats, Assets/Raymarching.mat:66-93
MATERIAL_FLOATS: Dict[str, float] = {
    "_ContactThreshold": 0.002, "_FractalRotationX": 0.19, "_FractalRotationY": -0.26, "_FractalRotationZ": 0.14,
    "_FractalScale": 2.0, "_Glossiness": 0.2, "_MaxDist": 50.0, "_MaxSteps": 400.0, "_NormalSampleScale": 0.00005,
    "_Reflections": 3.12, "_RefractiveIndex": 1.0, "_StepFactor": 0.7,
}

# Directional Light rotation, SampleScene.unity:458 (150 degrees about X): the light
# shines along its forward (0,-.5,-.8660254); _WorldSpaceLightPos0 points TOWARDS it.
LIGHT_ROTATION = np.array([0.9659258, 0.0, 0.0, 0.2588191], dtype=F)
WORLD_SPACE_LIGHT_POS0 = np.array([0.0, 0.5, 0.8660254, 0.0], dtype=F)


def default_camera(width: int = 1920, height: int = 1080) -> Camera:
    """Main Camera, SampleScene.unity:182-222."""
    return Camera(transform=Transform(position=vec3(2, 1, -11.01)), fieldOfView=80.0, aspect=width / height,
                  nearClipPlane=0.3, farClipPlane=1000.0, backgroundColor=(0.19215687, 0.3019608, 0.4745098, 0.0))


def default_transforms() -> List[Transform]:
    """"Position 0..3" in the order of the script's `transforms` array
    (SampleScene.unity:242-246; values :360, :489, :390, :139)."""
    return [Transform(position=vec3(-0.584, 0.049, 0.602)),      # sphere
            Transform(position=vec3(-0.74, 0.929, 0.0)),         # box
            Transform(position=vec3(-7.87, 0.0, 1.522)),         # link(5,5,.1)
            Transform(position=vec3(-3.074, -0.087, 0.08))]      # cylinder


def shadow_camera(width: int = 100, height: int = 100) -> Camera:
    """The (inactive) ShadowMap camera object, SampleScene.unity:287-289,316-318."""
    t = Transform(position=vec3(0, 3, 9),
                  rotation=np.array([-0.0000003870526, 0.9914449, -0.13052598, 0.0000029399619], dtype=F),
                  lossyScale=vec3(7, 4, 1))
    return Camera(transform=t, fieldOfView=60.0, aspect=width / height, nearClipPlane=0.3, farClipPlane=1000.0)


def light_camera(sw: int, sh: int) -> Camera:
    """a shadow camera 20 units from the objects towards the light (_WorldSpaceLightPos0), looking back at them: what a
    ShadowMap component placed at the light would render (Assets/ShadowMap.cs:50-56)"""
    centre = np.array([-3.0, 0.5, 0.5])
    L = WORLD_SPACE_LIGHT_POS0[:3].astype(np.float64)
    pos = centre + 20.0 * L
    return Camera(transform=Transform(position=pos.astype(F), rotation=quat_look_rotation(centre - pos)), fieldOfView=50.0,
                  aspect=sw / sh, nearClipPlane=0.3, farClipPlane=1000.0)


def light_rectangle() -> Transform:
    """a ShadowMapDirectional rectangle (Assets/ShadowMapDirectional.cs:25-34): 24 x 14 units, 20 units from the objects
    towards the light, facing them"""
    centre = np.array([-3.0, 0.5, 0.5])
    L = WORLD_SPACE_LIGHT_POS0[:3].astype(np.float64)
    pos = centre + 20.0 * L
    return Transform(position=pos.astype(F), rotation=quat_look_rotation(centre - pos), lossyScale=vec3(24, 14, 1))


def closeup_camera(centre, width: int, height: int) -> Camera:
    """a camera 3.4 units from `centre`, looking at it, fov 60: the close-up view of a fractal scene() body around slot 0"""
    centre = np.asarray(centre, dtype=np.float64)
    pos = centre + np.array([2.1, 1.3, -2.3])
    return Camera(transform=Transform(position=pos.astype(F), rotation=quat_look_rotation(centre - pos)),
                  fieldOfView=60.0, aspect=width / height, nearClipPlane=0.3, farClipPlane=1000.0)


def orbit_camera(k: int, n: int = 64, width: int = 7680, height: int = 4320) -> Camera:
    """Pose k of n on a circle of radius 11.5 about (-3, .5, .5) at height 1, looking
    at the centre; fov 80."""
    centre = np.array([-3.0, 0.5, 0.5])
    az = 2.0 * math.pi * k / n
    pos = np.array([centre[0] + 11.5 * math.sin(az), 1.0, centre[2] - 11.5 * math.cos(az)])
    rot = quat_look_rotation(centre - pos)
    return Camera(transform=Transform(position=pos.astype(F), rotation=rot), fieldOfView=80.0, aspect=width / height,
                  nearClipPlane=0.3, farClipPlane=1000.0)


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self) -> float:
        """[0,1) from the top 24 bits"""
        return (self.next() >> 40) / float(1 << 24)


def stress_primitives(n: int = 256, seed: int = 0x5EED0256) -> List[dict]:
    """N smooth-union primitives: type i mod 4 (sphere, box, link(.6,.4,.08), cylinder),
    position uniform in [-12,12] x [-.5,.5] x [-12,12], scale uniform in [.5,1.5]^3,
    uniform random unit quaternion, colour uniform in [.25,1]^3."""
    rng = SplitMix64(seed)
    params = [(1.0, 0, 0, 0), (1.0, 1.0, 1.0, 0), (0.6, 0.4, 0.08, 0), (0.5, 0.5, 0, 0)]
    out = []
    for i in range(n):
        u = [rng.uniform() for _ in range(13)]
        pos = [-12 + 24 * u[0], -0.5 + u[1], -12 + 24 * u[2]]
        scale = [0.5 + u[3], 0.5 + u[4], 0.5 + u[5]]
        # Shoemake's uniform unit quaternion
        s1, s2 = math.sqrt(1 - u[6]), math.sqrt(u[6])
        t1, t2 = 2 * math.pi * u[7], 2 * math.pi * u[8]
        quat = [s1 * math.sin(t1), s1 * math.cos(t1), s2 * math.sin(t2), s2 * math.cos(t2)]
        color = [0.25 + 0.75 * u[9], 0.25 + 0.75 * u[10], 0.25 + 0.75 * u[11]]
        out.append({"type": i % 4, "op": 0,
                    "pos": [float(F(v)) for v in pos], "quat": [float(F(v)) for v in quat],
                    "scale": [float(F(v)) for v in scale], "color": [float(F(v)) for v in color],
                    "params": params[i % 4]})
    return out


OP_SMAX, OP_NEGATE = 1, 2      # rm_primitive.op bits


def csg_primitives() -> List[dict]:
    """Seven primitives in front of the default camera that use every op: a box with a sphere
    carved out, a cylinder intersected with a sphere, a torus joined to a link, the whole clipped by
    a large sphere.  Tree (index order, odd element carried): (((0,1),(2,3)), ((4,5),6))."""
    q0 = [0.0, 0.0, 0.0, 1.0]
    q1 = [float(v) for v in quat_angle_axis(25.0, (0, 1, 0))]
    q2 = [float(v) for v in quat_angle_axis(-40.0, (1, 0, 1))]

    def prim(t, op, pos, quat, scale, color, params):
        return {"type": t, "op": op, "pos": [float(F(v)) for v in pos], "quat": [float(F(v)) for v in quat],
                "scale": [float(F(v)) for v in scale], "color": color, "params": params}
    return [
        prim(1, 0, (-1.4, 0.6, 0.0), q1, (1, 1, 1), (1.0, 0.4, 0.3), (2.0, 2.0, 2.0)),
        prim(0, OP_SMAX | OP_NEGATE, (-1.1, 0.9, -0.7), q0, (1, 1, 1), (0.3, 0.9, 1.0), (2.2,)),
        prim(3, 0, (-4.2, 0.2, 0.3), q2, (1.5, 1.2, 1.5), (0.9, 0.9, 0.3), (0.5, 0.5)),
        prim(0, OP_SMAX, (-4.2, 0.5, 0.3), q0, (1, 1, 1), (0.4, 1.0, 0.4), (1.9,)),
        prim(4, 0, (-7.0, 0.6, 1.0), q2, (2, 2, 2), (0.5, 0.5, 1.0), (0.5, 0.25)),
        prim(2, 0, (-7.87, 0.0, 1.522), q0, (1, 1, 1), (0.25, 0.25, 1.0), (5.0, 5.0, 0.1)),
        prim(0, OP_SMAX, (-4.0, 0.5, 0.5), q0, (1, 1, 1), (1.0, 1.0, 1.0), (13.0,)),
    ]


def csg_random_primitives(n: int, seed: int = 0x5EEDC56) -> List[dict]:
    """stress_primitives(n) with op bits that reach every kind of node of the tree walk while
    leaving something to see: primitives 0..7 stay a union (1 carved out of 0, 4 intersected with
    5); everything from 8 on is negated and joined by smax at the power-of-two boundaries (so
    it only carves), with further smax nodes at random (1 in 8) inside."""
    out = stress_primitives(n, seed)
    rng = SplitMix64(seed ^ 0xC56C56)
    for i, p in enumerate(out):
        u = rng.uniform()
        if i >= 8:
            boundary = (i & (i - 1)) == 0 or (i == 4 * (n // 4) and n % 4 != 0)
            p["op"] = OP_NEGATE | (OP_SMAX if (boundary or u < 0.125) else 0)
    if n > 1:
        out[1]["op"] = OP_SMAX | OP_NEGATE
    if n > 5:
        out[5]["op"] = OP_SMAX
    return out


def stress_camera(width: int = 1920, height: int = 1080) -> Camera:
    """Camera at (0, .75, -14) pitched 2 degrees down: grazing rays over the slab."""
    rot = quat_angle_axis(2.0, (1, 0, 0))
    return Camera(transform=Transform(position=vec3(0, 0.75, -14), rotation=rot), fieldOfView=80.0,
                  aspect=width / height, nearClipPlane=0.3, farClipPlane=1000.0)


STRESS_FLOATS = {"_MaxSteps": 512.0, "_MaxDist": 50.0, "_StepFactor": 0.7}
