"""Host-side counterparts of the reference's C# scripts (no GPU)."""
import math

import numpy as np
import pytest

from raymarching_amd import Material, Raymarching, ShadowMap, ShadowMapDirectional, get_frustum_corners, scenes
from raymarching_amd.unity import Camera, F, Matrix4x4, Transform, quat_angle_axis, quat_look_rotation, quat_rotate, vec3


def test_frustum_corners_rows():
    """Assets/Raymarching.cs:32-56: rows TL,TR,BR,BL of un-normalised rays with z = -1"""
    cam = scenes.default_camera(1920, 1080)
    m = get_frustum_corners(cam)
    t = math.tan(math.radians(40.0))
    a = t * 16 / 9
    rows = m.as_rows()
    np.testing.assert_allclose(rows[0], [-a, t, -1, 0], rtol=1e-6)
    np.testing.assert_allclose(rows[1], [a, t, -1, 0], rtol=1e-6)
    np.testing.assert_allclose(rows[2], [a, -t, -1, 0], rtol=1e-6)
    np.testing.assert_allclose(rows[3], [-a, -t, -1, 0], rtol=1e-6)
    assert rows[0][1] == pytest.approx(0.8390996, rel=1e-6) and rows[1][0] == pytest.approx(1.4917326, rel=1e-6)


def test_camera_to_world_is_trs_with_negated_z():
    cam = scenes.default_camera()
    rows = cam.cameraToWorldMatrix.as_rows()
    np.testing.assert_array_equal(rows, np.array([[1, 0, 0, 2], [0, 1, 0, 1], [0, 0, -1, -11.01], [0, 0, 0, 1]], dtype=F))
    cam2 = scenes.orbit_camera(16, 64)           # azimuth 90 deg: at +x of the centre, looking along -x
    fwd = cam2.transform.forward
    np.testing.assert_allclose(fwd, [-0.999, -0.0435, 0], atol=2e-3)
    rows2 = cam2.cameraToWorldMatrix.as_rows()
    np.testing.assert_allclose(rows2[:3, 2], -fwd, atol=1e-6)
    np.testing.assert_allclose(rows2[:3, 3], cam2.transform.position)


def test_get_positions_rotations_scales_pack_rows():
    """Assets/Raymarching.cs:58-83: at most 4 transforms, one per row, identity elsewhere"""
    ts = scenes.default_transforms()
    ts[1].rotation = quat_angle_axis(90, (0, 0, 1))
    ts[3].lossyScale = vec3(2, 3, 4)
    with Material() as mat:
        rm = Raymarching(scenes.default_camera(), mat, ts + [Transform(position=vec3(9, 9, 9))])
        P, R, S = rm.GetPositions(), rm.GetRotations(), rm.GetScales()
        np.testing.assert_allclose(P.GetRow(0), [-0.584, 0.049, 0.602, 0], rtol=1e-7)
        np.testing.assert_allclose(P.GetRow(3), [-3.074, -0.087, 0.08, 0], rtol=1e-7)
        np.testing.assert_allclose(R.GetRow(0), [0, 0, 0, 1])
        np.testing.assert_allclose(R.GetRow(1), [0, 0, math.sqrt(.5), math.sqrt(.5)], rtol=1e-6)
        np.testing.assert_allclose(S.GetRow(3), [2, 3, 4, 0])
        rm2 = Raymarching(scenes.default_camera(), mat, ts[:2])
        np.testing.assert_allclose(rm2.GetPositions().GetRow(2), [0, 0, 1, 0])       # untouched identity row
        rm.bind()
        assert mat.GetMatrix("_Scales").GetRow(3).tolist() == [2, 3, 4, 0]
        assert mat.GetVector("_WorldSpaceCameraPos")[:3].tolist() == pytest.approx([2, 1, -11.01])
        assert mat.GetVector("_ProjectionParams")[2] == 1000.0


def test_light_direction_from_scene_rotation():
    """_WorldSpaceLightPos0 = -forward of the Directional Light (SampleScene.unity:458)"""
    fwd = quat_rotate(scenes.LIGHT_ROTATION, vec3(0, 0, 1))
    np.testing.assert_allclose(-fwd, scenes.WORLD_SPACE_LIGHT_POS0[:3], atol=1e-6)


def test_shadowmap_directional_corners():
    """Assets/ShadowMapDirectional.cs:25-34"""
    t = Transform(position=vec3(1, 2, 3), rotation=quat_angle_axis(90, (0, 1, 0)), lossyScale=vec3(4, 2, 1))
    with Material() as mat:
        smd = ShadowMapDirectional(t, mat)
        smd.Update()
        # unit quad (+-.5,+-.5,0) scaled to 4 x 2, rotated 90 deg about y (x -> -z), translated
        np.testing.assert_allclose(smd.corners[0], [1, 3, 3 + 2], atol=1e-6)     # (-2, 1, 0) -> (0, 1, 2)
        np.testing.assert_allclose(smd.corners[1], [1, 3, 3 - 2], atol=1e-6)
        np.testing.assert_allclose(smd.corners[2], [1, 1, 3 - 2], atol=1e-6)
        np.testing.assert_allclose(smd.corners[3], [1, 1, 3 + 2], atol=1e-6)
        np.testing.assert_allclose(t.forward, [1, 0, 0], atol=1e-6)
        with pytest.raises(ValueError):
            smd.Render()


def test_shadowmap_frustum_uses_its_own_camera():
    cam = scenes.shadow_camera(100, 100)
    with Material("Hidden/ShadowMap") as mat:
        sm = ShadowMap(cam, mat)
        rows = sm.GetFrustumCorners().as_rows()
        t = math.tan(math.radians(30.0))
        np.testing.assert_allclose(rows[1], [t, t, -1, 0], rtol=1e-6)


def test_look_rotation_and_quat_rotate_agree():
    for d in [(0, 0, 1), (1, 0, 0), (0.3, -0.2, -0.9), (-1, 0.5, 0.1)]:
        q = quat_look_rotation(d)
        f = quat_rotate(q, vec3(0, 0, 1))
        np.testing.assert_allclose(f, np.asarray(d) / np.linalg.norm(d), atol=1e-6)


def test_stress_scene_is_deterministic():
    a = scenes.stress_primitives(256)
    b = scenes.stress_primitives(256)
    assert a == b and len(a) == 256
    assert [p["type"] for p in a[:8]] == [0, 1, 2, 3, 0, 1, 2, 3]
    pos = np.array([p["pos"] for p in a])
    assert pos[:, 0].min() >= -12 and pos[:, 0].max() <= 12 and np.abs(pos[:, 1]).max() <= 0.5
    q = np.array([p["quat"] for p in a])
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
    r = scenes.SplitMix64(0)
    assert r.next() == 0xE220A8397B1DCDAF            # published splitmix64 test vector for seed 0
