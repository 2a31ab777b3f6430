"""Minimal float32 stand-ins for the UnityEngine value types the reference's C#
scripts use (Vector3, Quaternion, Matrix4x4, Transform, Camera).

Only what Assets/Raymarching.cs, Assets/ShadowMap.cs and
Assets/ShadowMapDirectional.cs touch is modelled.  All arithmetic is done in
numpy.float32 so that the matrices handed to the plugin are the ones Unity's
own float math would produce.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

F = np.float32
DEG2RAD = F(math.pi * 2.0 / 360.0)          # Mathf.Deg2Rad


def vec3(x, y, z) -> np.ndarray:
    return np.array([x, y, z], dtype=F)


def quat_identity() -> np.ndarray:
    return np.array([0, 0, 0, 1], dtype=F)


def quat_mul(a, b) -> np.ndarray:
    """Hamilton product of (x,y,z,w) quaternions (Unity's operator*)."""
    ax, ay, az, aw = [F(v) for v in a]
    bx, by, bz, bw = [F(v) for v in b]
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz], dtype=F)


def quat_rotate(q, v) -> np.ndarray:
    """Quaternion * Vector3 with Unity's expansion (rotation matrix from doubled products)."""
    x, y, z, w = [F(c) for c in q]
    x2, y2, z2 = x * F(2), y * F(2), z * F(2)
    xx, yy, zz = x * x2, y * y2, z * z2
    xy, xz, yz = x * y2, x * z2, y * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    vx, vy, vz = [F(c) for c in v]
    return np.array([
        (F(1) - (yy + zz)) * vx + (xy - wz) * vy + (xz + wy) * vz,
        (xy + wz) * vx + (F(1) - (xx + zz)) * vy + (yz - wx) * vz,
        (xz - wy) * vx + (yz + wx) * vy + (F(1) - (xx + yy)) * vz], dtype=F)


def quat_angle_axis(deg: float, axis) -> np.ndarray:
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    h = math.radians(deg) * 0.5
    s = math.sin(h)
    return np.array([a[0] * s, a[1] * s, a[2] * s, math.cos(h)], dtype=F)


def quat_look_rotation(forward, up=(0, 1, 0)) -> np.ndarray:
    """Quaternion.LookRotation: +Z along `forward`, +Y as close to `up` as possible."""
    f = np.asarray(forward, dtype=np.float64)
    f = f / np.linalg.norm(f)
    r = np.cross(np.asarray(up, dtype=np.float64), f)
    r = r / np.linalg.norm(r)
    u = np.cross(f, r)
    m = np.stack([r, u, f], axis=1)            # columns = right, up, forward
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    return np.array(q, dtype=F)


class Matrix4x4:
    """Blittable 4x4 float matrix; `.m` is the 16 floats in Unity's field order
    m00,m10,m20,m30,m01,... (column-major)."""

    def __init__(self, m=None):
        self.m = np.zeros(16, dtype=F) if m is None else np.asarray(m, dtype=F).reshape(16).copy()

    @staticmethod
    def identity() -> "Matrix4x4":
        r = Matrix4x4()
        r.m[[0, 5, 10, 15]] = 1
        return r

    def SetRow(self, i: int, v) -> None:
        """Vector3 rows are widened to Vector4 with w = 0 as Unity's implicit conversion does."""
        v = np.asarray(v, dtype=F).reshape(-1)
        if v.size == 3:
            v = np.append(v, F(0))
        self.m[i::4] = v

    def GetRow(self, i: int) -> np.ndarray:
        return self.m[i::4].copy()

    def as_rows(self) -> np.ndarray:
        return self.m.reshape(4, 4).T.copy()

    @staticmethod
    def from_rows(rows) -> "Matrix4x4":
        return Matrix4x4(np.asarray(rows, dtype=F).reshape(4, 4).T.reshape(16))

    def __mul__(self, other: "Matrix4x4") -> "Matrix4x4":
        """Matrix4x4 operator*: the matrix product, accumulated in float32 left to right"""
        a, b = self.as_rows(), other.as_rows()
        out = np.zeros((4, 4), dtype=F)
        for i in range(4):
            for j in range(4):
                acc = F(0)
                for k in range(4):
                    acc = F(acc + a[i, k] * b[k, j])
                out[i, j] = acc
        return Matrix4x4.from_rows(out)

    @staticmethod
    def TRS(pos, q, scale) -> "Matrix4x4":
        cols = [quat_rotate(q, vec3(1, 0, 0)) * F(scale[0]),
                quat_rotate(q, vec3(0, 1, 0)) * F(scale[1]),
                quat_rotate(q, vec3(0, 0, 1)) * F(scale[2])]
        rows = np.zeros((4, 4), dtype=F)
        for c in range(3):
            rows[0:3, c] = cols[c]
        rows[0:3, 3] = np.asarray(pos, dtype=F)
        rows[3, 3] = 1
        return Matrix4x4.from_rows(rows)


@dataclass
class Transform:
    position: np.ndarray = field(default_factory=lambda: vec3(0, 0, 0))
    rotation: np.ndarray = field(default_factory=quat_identity)
    lossyScale: np.ndarray = field(default_factory=lambda: vec3(1, 1, 1))

    def __post_init__(self):
        self.position = np.asarray(self.position, dtype=F)
        self.rotation = np.asarray(self.rotation, dtype=F)
        self.lossyScale = np.asarray(self.lossyScale, dtype=F)

    @property
    def forward(self) -> np.ndarray:
        return quat_rotate(self.rotation, vec3(0, 0, 1))

    def TransformDirection(self, v) -> np.ndarray:
        return quat_rotate(self.rotation, v)


@dataclass
class Camera:
    transform: Transform = field(default_factory=Transform)
    fieldOfView: float = 60.0
    aspect: float = 16.0 / 9.0
    nearClipPlane: float = 0.3
    farClipPlane: float = 1000.0
    backgroundColor: tuple = (0.19215687, 0.3019608, 0.4745098, 0.0)

    @property
    def cameraToWorldMatrix(self) -> Matrix4x4:
        """TRS(position, rotation, 1) with the Z column negated: Unity's camera
        space looks down -Z (hence -Vector3.forward at Assets/Raymarching.cs:45-48)."""
        m = Matrix4x4.TRS(self.transform.position, self.transform.rotation, (1, 1, 1))
        rows = m.as_rows()
        rows[:, 2] = -rows[:, 2]
        return Matrix4x4.from_rows(rows)

    @property
    def worldToCameraMatrix(self) -> Matrix4x4:
        """inverse of cameraToWorldMatrix (a rotation, a translation and the Z flip): R^T and -R^T t"""
        rows = self.cameraToWorldMatrix.as_rows().astype(np.float64)
        r, t = rows[:3, :3], rows[:3, 3]
        out = np.zeros((4, 4), dtype=np.float64)
        out[:3, :3] = r.T
        out[:3, 3] = -(r.T @ t)
        out[3, 3] = 1.0
        return Matrix4x4.from_rows(out.astype(F))

    @property
    def projectionMatrix(self) -> Matrix4x4:
        """Camera.projectionMatrix: OpenGL-convention perspective matrix (clip.w = -z_view = eye depth)"""
        t = math.tan(math.radians(float(self.fieldOfView)) * 0.5)
        n, fa = float(self.nearClipPlane), float(self.farClipPlane)
        rows = [[1.0 / (t * float(self.aspect)), 0, 0, 0], [0, 1.0 / t, 0, 0],
                [0, 0, -(fa + n) / (fa - n), -2.0 * fa * n / (fa - n)], [0, 0, -1, 0]]
        return Matrix4x4.from_rows(np.asarray(rows, dtype=F))
