"""Known-answer tests that follow from the closed forms of the reference's SDFs
(Assets/Raymarching.cginc).  The reference ships no test vectors, so these pin the
oracle to the formulas themselves."""
import math

import numpy as np
import pytest

from oracle.rm_oracle import RmoUniforms, Uniforms, f32, fp

F = np.float32


def p3(x, y, z):
    return fp(f32([x, y, z]))


def test_sphere(oracle):                       # cginc:169-171
    l = oracle.lib()
    assert l.rmo_sphere(p3(1, 0, 0), 1.0) == 0.5
    assert l.rmo_sphere(p3(0, 0, 0), 1.0) == -0.5
    assert l.rmo_sphere(p3(0, 3, 4), 2.0) == 4.0


def test_box(oracle):                          # cginc:173-177
    l = oracle.lib()
    b = fp(f32([1, 1, 1]))
    assert l.rmo_box(p3(1, 0, 0), b) == 0.5
    assert l.rmo_box(p3(0, 0, 0), b) == -0.5
    assert l.rmo_box(p3(0.5, 0.5, 0.5), b) == 0.0
    assert l.rmo_box(p3(3.5, 4.5, 0), b) == 5.0            # corner distance 3-4-5


def test_cylinder(oracle):                     # cginc:189-192: h is the radius, r the half height
    l = oracle.lib()
    assert l.rmo_cylinder(p3(1, 0, 0), .5, .5) == 0.5
    assert l.rmo_cylinder(p3(0, 1, 0), .5, .5) == 0.5
    assert l.rmo_cylinder(p3(0, 0, 0), .5, .5) == -0.5
    assert l.rmo_cylinder(p3(0, 0, 2), .5, .25) == 1.5


def test_link(oracle):                         # cginc:184-187 with (5,5,.1)
    l = oracle.lib()
    assert l.rmo_link(p3(0, 0, 0), 5, 5, .1) == pytest.approx(4.9, abs=1e-6)
    assert l.rmo_link(p3(5, 0, 0), 5, 5, .1) == pytest.approx(-0.1, abs=1e-7)
    assert l.rmo_link(p3(0, 10, 0), 5, 5, .1) == pytest.approx(-0.1, abs=1e-7)   # top of the rounded end
    assert l.rmo_link(p3(5, 3, 2), 5, 5, .1) == pytest.approx(1.9, abs=1e-6)     # straight section


def test_torus(oracle):                        # cginc:179-182
    l = oracle.lib()
    assert l.rmo_torus(p3(.5, 0, 0), .5, .25) == -0.25
    assert l.rmo_torus(p3(0, 0, 0), .5, .25) == 0.25


def test_mod_and_shmod(oracle):                # cginc:84-99
    l = oracle.lib()
    assert l.rmo_mod(5.5, 2.0) == 1.5
    assert l.rmo_mod(-0.5, 2.0) == 1.5           # made non-negative
    assert l.rmo_mod(-4.0, 2.0) == 0.0
    assert l.rmo_shmod(0.25, 2.0) == 0.25        # inside the centred cell: unchanged
    assert l.rmo_shmod(1.25, 2.0) == -0.75       # wraps to [-m/2, m/2)
    assert l.rmo_shmod(-1.0, 2.0) == -1.0
    assert l.rmo_shmod(1.0, 2.0) == -1.0
    assert l.rmo_shmod(7.0, 2.0) == -1.0 and l.rmo_shmod(8.5, 2.0) == 0.5


def test_folds(oracle):                        # cginc:156-165
    l = oracle.lib()
    out = f32([0, 0, 0])
    l.rmo_boxfold(p3(0.25, -0.75, 2.0), 1.0, fp(out))      # b = .5: inside unchanged, outside mirrored
    assert out.tolist() == [0.25, -0.25, -1.0]
    l.rmo_spherefold(p3(0.5, 0.0, 0.0), 1.0, fp(out))      # inversion in the unit sphere
    assert out.tolist() == [2.0, 0.0, 0.0]
    l.rmo_spherefold(p3(0.0, 3.0, 0.0), 2.0, fp(out))      # outside R: unchanged
    assert out.tolist() == [0.0, 3.0, 0.0]
    l.rmo_spherefold(p3(0.0, 0.0, -1.0), 2.0, fp(out))     # p * R * R / (r * r)
    assert out.tolist() == [0.0, 0.0, -4.0]


def test_smin(oracle):                         # cginc:110-113
    l = oracle.lib()
    out = f32([0, 0, 0, 0])

    def w(a, b):
        l.rmo_smin4(fp(f32([1, 0, 0, a])), fp(f32([0, 1, 0, b])), 0.5, fp(out))
        return float(out[3])
    assert w(0, 1) == 0.0
    assert w(1, 0) == 0.0
    assert w(0, 0) == -0.125
    assert w(0.25, 0.25) == 0.125
    # colour: far apart -> the nearer operand's colour, renormalised to its own length
    l.rmo_smin4(fp(f32([1, .25, .25, 0])), fp(f32([.25, 1, .25, 5])), 0.5, fp(out))
    np.testing.assert_allclose(out[:3], [1, .25, .25], rtol=3e-7)
    # equal distances: h = .5, colour = normalize(mid) * mean length
    l.rmo_smin4(fp(f32([1, 0, 0, 1])), fp(f32([0, 1, 0, 1])), 0.5, fp(out))
    np.testing.assert_allclose(out[:3], [math.sqrt(.5), math.sqrt(.5), 0], rtol=3e-7)


def test_smax_closed_forms(oracle):             # cginc:120-123
    l = oracle.lib()
    out = f32([0, 0, 0, 0])

    def w(a, b):
        l.rmo_smax4(fp(f32([1, 0, 0, a])), fp(f32([0, 1, 0, b])), 0.5, fp(out))
        return float(out[3])
    assert w(0, 1) == 1.0                      # far apart: the larger operand
    assert w(1, 0) == 1.0
    assert w(0, 0) == -0.125                   # the reference subtracts k*h*(1-h) here too
    # colour is a plain lerp(b.rgb, a.rgb, h): h = 1 when a is the larger
    l.rmo_smax4(fp(f32([1, .25, .25, 5])), fp(f32([.25, 1, .25, 0])), 0.5, fp(out))
    assert out[:3].tolist() == [1, .25, .25]
    l.rmo_smax4(fp(f32([1, 0, 0, 1])), fp(f32([0, 1, 0, 1])), 0.5, fp(out))
    assert out[:3].tolist() == [.5, .5, 0]


def test_csg_ops_on_a_table_scene(oracle, default_setup):
    """box with a negated sphere joined by smax = box minus sphere: the sphere's interior is outside"""
    _, _, props = default_setup(16, 9)
    box = {"type": 1, "op": 0, "pos": (0, 0, 0), "quat": (0, 0, 0, 1), "scale": (1, 1, 1), "color": (1, 1, 1), "params": (2, 2, 2)}
    hole = {"type": 0, "op": 3, "pos": (0, 0, 0), "quat": (0, 0, 0, 1), "scale": (1, 1, 1), "color": (1, 1, 1), "params": (1,)}
    l = oracle.lib()
    out = f32([0, 0, 0, 0])
    uni = oracle.Uniforms(props, prims=[box, hole])
    l.rmo_scene(uni.ref, p3(0, 0, 0), fp(out))
    assert out[3] == 0.5                       # centre: inside the box (-1) but 0.5 inside the carved sphere
    l.rmo_scene(uni.ref, p3(0.9, 0.9, 0.9), fp(out))
    assert out[3] < 0                          # box corner region, outside the sphere: solid
    uni = oracle.Uniforms(props, prims=[box, dict(hole, op=1)])          # plain intersection
    l.rmo_scene(uni.ref, p3(0, 0, 0), fp(out))
    assert out[3] == -0.5
    l.rmo_scene(uni.ref, p3(0.9, 0.9, 0.9), fp(out))
    assert out[3] > 0


def test_fractal_bodies_closed_forms(oracle, default_setup):      # cginc:264-313
    _, _, props = default_setup(16, 9)
    props = dict(props, _FractalRotationX=0.0, _FractalRotationY=0.0, _FractalRotationZ=0.0, _FractalScale=2.0)
    uni = oracle.Uniforms(props)
    l = oracle.lib()
    out = f32([0, 0, 0, 0])
    # sierpinski outside the bailout radius: the loop does not run, (sqrt(r) - 2) * scale^0, traps stay at 20
    l.rmo_sierpinski(uni.ref, p3(5, 0, 0), fp(out))
    assert out.tolist() == [20, 20, 20, 3]
    # one fold: (1,1,1) -> 2*(1,1,1) - 1 = (1,1,1) is the fixed point of the map; after 15 rounds
    # (sqrt(3) - 2) * 2^-15, traps = distance from (1,1,1) to the unit points = sqrt(2)
    l.rmo_sierpinski(uni.ref, p3(1, 1, 1), fp(out))
    s3, s2 = np.sqrt(np.float32(3)), np.sqrt(np.float32(2))
    assert out.tolist() == [s2, s2, s2, (s3 - np.float32(2)) * np.float32(2.0 ** -15)]
    # menger without rotation on integer points stays in exactly representable integers
    x, y, z = 0.0, 0.0, 0.0
    tr = [50.0, 50.0, 50.0]
    for _ in range(12):
        x, y, z = sorted((abs(x), abs(y), abs(z)), reverse=True)
        x, y, z = 3 * x - 2, 3 * y - 2, 3 * z - 2
        if z < -1:
            z += 2
        tr = [min(tr[0], math.sqrt(x * x + y * y + z * z)), min(tr[1], math.sqrt(x * x + (y - .5) ** 2 + z * z)),
              min(tr[2], math.sqrt(x * x + y * y + (z - .5) ** 2))]
    l.rmo_menger(uni.ref, p3(0, 0, 0), fp(out))
    want_w = (np.sqrt(np.float32(x * x + y * y + z * z)) - np.float32(1.5)) * np.float32(3.0 ** -12)
    assert out[3] == want_w
    np.testing.assert_allclose(out[:3], tr, rtol=1e-6)


def test_canonical_transcendentals_are_faithful(oracle):
    """The double-evaluated sin/cos/acos/asin/atan2/log/pow that the mandelbulb bodies are defined
    with: each must be the float nearest to libm's double result on a broad sample, and the two
    restatements must agree bit for bit."""
    from oracle import numpy_restatement as NP
    l = oracle.lib()
    rng = np.random.default_rng(7)

    def same(got, want):
        got, want = np.asarray(got, dtype=np.float32), np.asarray(want, dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))

    x = rng.uniform(-70, 70, 4000).astype(np.float32)
    same([l.rmo_c_sin(float(v)) for v in x], np.sin(x.astype(np.float64)))
    same([l.rmo_c_cos(float(v)) for v in x], np.cos(x.astype(np.float64)))
    same(NP.c_sin(x), np.sin(x.astype(np.float64)))
    same(NP.c_cos(x), np.cos(x.astype(np.float64)))
    v = rng.uniform(-1, 1, 4000).astype(np.float32)
    v[:5] = [1, -1, 0, 0.5, -0.5]
    same([l.rmo_c_acos(float(t)) for t in v], np.arccos(v.astype(np.float64)))
    same([l.rmo_c_asin(float(t)) for t in v], np.arcsin(v.astype(np.float64)))
    same(NP.c_acos(v), np.arccos(v.astype(np.float64)))
    same(NP.c_asin(v), np.arcsin(v.astype(np.float64)))
    y, xx = rng.normal(size=4000).astype(np.float32), rng.normal(size=4000).astype(np.float32)
    y[:4], xx[:4] = [0, 1, -1, 0], [-1, 0, 0, 1]
    same([l.rmo_c_atan2(float(a), float(b)) for a, b in zip(y, xx)], np.arctan2(y.astype(np.float64), xx.astype(np.float64)))
    same(NP.c_atan2(y, xx), np.arctan2(y.astype(np.float64), xx.astype(np.float64)))
    r = np.exp(rng.uniform(-30, 5, 4000)).astype(np.float32)
    same([l.rmo_c_log(float(t)) for t in r], np.log(r.astype(np.float64)))
    same(NP.c_log(r), np.log(r.astype(np.float64)))
    for e in (7.0, 6.0, 2.0, 1.0, float(np.float32(3.7))):
        same([l.rmo_c_pow(float(t), e) for t in r], np.power(r.astype(np.float64), e))
        same(NP.c_pow(r, e), np.power(r.astype(np.float64), e))
    assert l.rmo_c_pow(0.0, 7.0) == 0.0 and l.rmo_c_pow(0.0, 0.0) == 1.0 and l.rmo_c_log(0.0) == -math.inf
    assert math.isnan(l.rmo_c_atan2(0.0, 0.0)) and math.isnan(l.rmo_c_log(-1.0))


def test_mandelbulb_outside_the_bailout_radius(oracle):        # cginc:194-227
    """|p| > 10: the loop leaves on its first test; .5*log(r)*r/1, traps = distances of p itself"""
    l = oracle.lib()
    out = f32([0, 0, 0, 0])
    l.rmo_mandelbulb(p3(0, 0, 16), 7.0, fp(out))
    assert out[3] == np.float32(0.5) * np.float32(math.log(16.0)) * np.float32(16)
    assert out[:3].tolist() == [10, 10, 10]
    l.rmo_mandelbrot(p3(12, 0, 0), 2.0, fp(out))
    assert out[3] == np.float32(0.5) * np.float32(math.log(12.0)) * np.float32(12)
    assert out[:3].tolist() == [10, 10, 10]


def test_rotate_and_remap(oracle):             # cginc:62-66, 80-82
    l = oracle.lib()
    out = f32([0, 0, 0])
    l.rmo_rotate_q(p3(1, 2, 3), fp(f32([0, 0, 0, 1])), fp(out))
    assert out.tolist() == [1, 2, 3]
    s = math.sqrt(.5)
    l.rmo_rotate_q(p3(1, 0, 0), fp(f32([0, 0, s, s])), fp(out))          # +90 deg about z, applied as the conjugate
    np.testing.assert_allclose(out, [0, -1, 0], atol=1e-7)
    assert l.rmo_remap(0, 0, 1, .25, 1) == 0.25
    assert l.rmo_remap(1, 0, 1, .25, 1) == 1.0
    assert l.rmo_remap(.5, 0, 1, .25, 1) == 0.625


def test_vert_corner_index(oracle):            # cginc:30 with Raymarching.cs:50-53
    l = oracle.lib()
    assert l.rmo_vert_corner_index(0, 1) == 0      # top left
    assert l.rmo_vert_corner_index(1, 1) == 1      # top right
    assert l.rmo_vert_corner_index(1, 0) == 2      # bottom right
    assert l.rmo_vert_corner_index(0, 0) == 3      # bottom left


def test_pixel_rays_and_centre_miss(oracle, default_setup):
    _, _, props = default_setup(640, 360)
    uni = Uniforms(props)
    l = oracle.lib()
    rd = f32([0, 0, 0])
    # un-normalised rays with world z = +1 for the identity camera: dm is linear eye depth
    for (x, y) in [(0, 0), (639, 359), (320, 180), (17, 300)]:
        l.rmo_pixel_ray(uni.ref, x, y, 640, 360, fp(rd))
        assert rd[2] == 1.0
    t = math.tan(math.radians(40.0))
    l.rmo_pixel_ray(uni.ref, 0, 0, 640, 360, fp(rd))
    np.testing.assert_allclose(rd[:2], [-t * 16 / 9 * (1 - 1 / 640), -t * (1 - 1 / 360)], rtol=2e-6)
    # the centre-of-screen ray points down +z and misses (the geometry is left of centre)
    ro = f32(props["_WorldSpaceCameraPos"][:3])
    length = f32([0])
    import ctypes as C
    ev = C.c_int()
    hit = l.rmo_raymarch(uni.ref, fp(ro), fp(f32([0, 0, 1])), 1000.0, fp(length), C.byref(ev))
    assert hit == 0 and length[0] > 50.0
    # a ray aimed at the sphere hits it at its front surface: centre distance minus radius
    c = f32([-0.584, 0.049, 0.602])
    d = (c - ro) / (c - ro)[2]
    hit = l.rmo_raymarch(uni.ref, fp(ro), fp(f32(d)), 1000.0, fp(length), C.byref(ev))
    assert hit == 1
    dist = float(length[0]) * float(np.linalg.norm(d))
    assert dist == pytest.approx(float(np.linalg.norm(c - ro)) - 0.5, rel=5e-3)


def test_scene_is_smin_tree(oracle, default_setup):
    """scene(p).w == smin(smin(o0,o1), smin(o2,o3)) built from the primitive probes (cginc:363)."""
    _, _, props = default_setup(64, 36)
    uni = Uniforms(props)
    l = oracle.lib()
    rng = np.random.default_rng(1)
    pos = np.asarray(props["_Positions"], dtype=F).reshape(4, 4).T
    out, s01, s23, s = f32([0] * 4), f32([0] * 4), f32([0] * 4), f32([0] * 4)
    cols = [(1, .25, .25), (.25, 1, .25), (.25, .25, 1), (1, 1, .25)]
    for _ in range(200):
        p = f32(rng.uniform([-13, -6, -2], [1, 6, 3]))
        q = [f32(p - pos[i, :3]) for i in range(4)]
        o = [l.rmo_sphere(fp(q[0]), 1.0), l.rmo_box(fp(q[1]), fp(f32([1, 1, 1]))),
             l.rmo_link(fp(q[2]), 5, 5, .1), l.rmo_cylinder(fp(q[3]), .5, .5)]
        v = [f32(list(cols[i]) + [o[i]]) for i in range(4)]
        l.rmo_smin4(fp(v[0]), fp(v[1]), .5, fp(s01))
        l.rmo_smin4(fp(v[2]), fp(v[3]), .5, fp(s23))
        l.rmo_smin4(fp(s01), fp(s23), .5, fp(s))
        l.rmo_scene(uni.ref, fp(p), fp(out))
        assert out.tobytes() == s.tobytes()


def test_unorm8_roundtrip_of_source_bytes(oracle, default_setup):
    """tex = byte/255 shown on a miss pixel quantises back to the same byte."""
    mat, _, props = default_setup(256, 1)
    # camera looking away from everything: all pixels miss
    props = dict(props)
    props["_WorldSpaceCameraPos"] = f32([0, 500, 0, 0])
    src = np.zeros((1, 256, 4), dtype=np.uint8)
    src[0, :, 0] = np.arange(256)
    src[0, :, 1] = np.arange(256)[::-1]
    src[0, :, 2] = 7
    src[0, :, 3] = 3
    out = oracle.render(Uniforms(props), 256, 1, src=src)
    assert out["stats"]["n_hit"] == 0
    assert np.array_equal(out["rgba"][..., :3], src[..., :3])
    assert (out["rgba"][..., 3] == 255).all()


def test_rand_noise_terrain_closed_forms(oracle):          # cginc:125-149, 315-330
    from oracle import numpy_restatement as NP
    l = oracle.lib()
    f2 = lambda x, y: fp(f32([x, y]))
    # rand = frac(sin(dot(co, (12.9898, 78.233))) * 43758.5453) in [0, 1); rand((0,0)) = frac(sin(0) * c) = 0
    assert l.rmo_rand(f2(0, 0)) == 0.0
    for co in [(1, 0), (0, 1), (3, -7), (-12, 5), (101, 17)]:
        want = math.sin(float(F(co[0]) * F(12.9898) + F(co[1]) * F(78.233)))
        want = F(F(want) * F(43758.5453))
        want = float(want - np.floor(want))
        got = l.rmo_rand(f2(*co))
        assert 0.0 <= got < 1.0 and got == pytest.approx(want, abs=1e-6)
        assert got == float(NP.rand2(np.array([co[0]], F), np.array([co[1]], F))[0])
    # noise at lattice points is the lattice value (f = 0 => u = 0); at the centre of a cell the mean of the corners
    for i, j in [(0, 0), (2, 3), (-4, 1)]:
        assert l.rmo_noise(f2(i, j)) == l.rmo_rand(f2(i, j))
        a, b = l.rmo_rand(f2(i, j)), l.rmo_rand(f2(i + 1, j))
        c, d = l.rmo_rand(f2(i, j + 1)), l.rmo_rand(f2(i + 1, j + 1))
        assert l.rmo_noise(f2(i + .5, j + .5)) == pytest.approx((a + b + c + d) / 4, abs=2e-6)
        assert l.rmo_noise(f2(i + .5, j)) == pytest.approx((a + b) / 2, abs=2e-6)
    # terrain(0): p = 0 in all three octaves, noise(0) = rand(0) = 0 -> height 0
    assert l.rmo_terrain(f2(0, 0)) == 0.0
    # a general point, against the formula written out with the probes above: x = (1e5, 2e5) -> p = (1.2, 2.4) (float32)
    x = np.array([1e5, 2e5], F)
    p = x * F(0.003) / F(250)
    a, b, dsum = F(0), F(1), F(0)
    for _ in range(3):
        n = F(l.rmo_noise(fp(p.copy())))
        dsum = F(dsum + n)
        a = F(a + F(b * n) / F(F(1) + F(F(dsum * dsum) + F(0))))
        b = F(b * F(.5))
        p = np.array([F(F(F(.8) * p[0]) * F(2)), F(F(F(-.6) * p[1]) * F(2))], F)
    assert l.rmo_terrain(fp(x)) == float(F(F(30000.0) * a))
    xs, ys = np.array([1e5, -3.3e4, 7.0], F), np.array([2e5, 9.1e4, -2.0], F)
    got = NP.terrain_w(xs, ys)
    assert [l.rmo_terrain(f2(float(u), float(v))) for u, v in zip(xs, ys)] == [float(g) for g in got]


def test_light_helpers_closed_forms(oracle):               # shader:84-100, 129-147
    l = oracle.lib()
    lp = p3(0, 10, 0)
    # light straight above a point whose normal points at it: dot(n, l - p) = 10 -> clamp -> 1
    assert l.rmo_getbrightness(p3(0, 0, 0), p3(0, 1, 0), lp, 1) == pytest.approx(1.0 * (1 - 10 / 1000), rel=1e-6)      # hard: lerp(1, 0, dist/range)
    assert l.rmo_getbrightness(p3(0, 0, 0), p3(0, 1, 0), lp, 0) == pytest.approx(1.0 * (1 - 10 / 1000), rel=1e-6)      # soft: (1*.5+.5)^2 = 1
    assert l.rmo_getbrightness(p3(0, 0, 0), p3(0, -1, 0), lp, 1) == 0.0                                               # facing away
    assert l.rmo_getbrightness(p3(0, 0, 0), p3(0, -1, 0), lp, 0) == 0.0                                               # (-1*.5+.5)^2 = 0
    assert l.rmo_getbrightness(p3(0, 0, 0), p3(1, 0, 0), lp, 0) == pytest.approx(0.25 * 0.99, rel=1e-6)               # perpendicular: (.5)^2
    assert l.rmo_getbrightness(p3(0, -990, 0), p3(0, 1, 0), lp, 0) == 0.0                                             # dist >= range
    # getscatter: a ray along x passing the light at distance h: s = 1/h, integral = s * (atan((L + b) s) - atan(b s))
    h, L = 2.0, 8.0
    got = l.rmo_getscatter(p3(-3, 0, 0), p3(1, 0, 0), L, p3(0, h, 0))
    want = (1 / h) * (math.atan((L - 3) / h) - math.atan(-3 / h))
    assert got == pytest.approx(want, rel=2e-6)
    assert l.rmo_c_atan(1.0) == float(F(math.pi / 4)) and l.rmo_c_atan(0.0) == 0.0
    # refract2(i, n, eta): eta' = 2 - eta; i*eta' - n*(-cosi + eta'*cosi); eta = 1 leaves i unchanged
    out = f32([0, 0, 0])
    l.rmo_refract2(p3(0.6, -0.8, 0), p3(0, 1, 0), 1.0, fp(out))
    assert out.tolist() == pytest.approx([0.6, -0.8, 0.0], abs=1e-7)
    l.rmo_refract2(p3(0.6, -0.8, 0), p3(0, 1, 0), 1.5, fp(out))          # eta' = .5, cosi = -.8: i*.5 - n*(.8 - .4)
    assert out.tolist() == pytest.approx([0.3, -0.8, 0.0], abs=1e-7)


def test_scalar_smin_and_smax(oracle):          # cginc:105-108, 115-118
    """the scalar overloads nothing calls: smin is the .w lane of the float4 one; smax as WRITTEN returns lerp(-b, -a, h)
    without the outer negation the float4 overload has, i.e. minus the larger operand once they are k apart"""
    l = oracle.lib()
    assert l.rmo_smin1(0.0, 1.0, 0.5) == 0.0 and l.rmo_smin1(1.0, 0.0, 0.5) == 0.0
    assert l.rmo_smin1(0.0, 0.0, 0.5) == -0.125 and l.rmo_smin1(0.25, 0.25, 0.5) == 0.125
    out = f32([0, 0, 0, 0])
    rng = np.random.default_rng(5)
    for a, b in rng.uniform(-2, 2, (200, 2)).astype(np.float32):
        l.rmo_smin4(fp(f32([1, 0, 0, a])), fp(f32([0, 1, 0, b])), 0.5, fp(out))
        assert l.rmo_smin1(float(a), float(b), 0.5) == float(out[3])
    assert l.rmo_smax1(1.0, 0.0, 0.5) == -1.0          # h = 1: lerp(-b, -a, 1) = -a
    assert l.rmo_smax1(0.0, 1.0, 0.5) == -1.0          # h = 0: -b
    assert l.rmo_smax1(0.0, 0.0, 0.5) == -0.125
    for a, b in rng.uniform(-2, 2, (200, 2)).astype(np.float32):
        l.rmo_smax4(fp(f32([1, 0, 0, a])), fp(f32([0, 1, 0, b])), 0.5, fp(out))
        h = np.float32(min(max(np.float32(0.5) + np.float32(0.5) * (a - b) / np.float32(0.5), 0), 1))
        blend = np.float32(0.5) * h * (np.float32(1) - h)
        # float4: -lerp(-b,-a,h) - k*h*(1-h); scalar: lerp(-b,-a,h) - k*h*(1-h): their sum is -2*k*h*(1-h)
        np.testing.assert_allclose(l.rmo_smax1(float(a), float(b), 0.5) + float(out[3]), -2.0 * float(blend), atol=1e-6)


def test_sq_and_luminance(oracle):              # cginc:68-74, 151-154
    l = oracle.lib()
    assert l.rmo_sq(3.0) == 9.0 and l.rmo_sq(-0.5) == 0.25
    out = f32([0, 0, 0])
    l.rmo_sq3(p3(1.5, -2.0, 0.0), fp(out))
    assert out.tolist() == [2.25, 4.0, 0.0]
    assert l.rmo_luminance(p3(1, 0, 0)) == float(np.float32(0.2990))
    assert l.rmo_luminance(p3(0, 1, 0)) == float(np.float32(0.5870))
    assert l.rmo_luminance(p3(0, 0, 1)) == float(np.float32(0.1140))
    np.testing.assert_allclose(l.rmo_luminance(p3(1, 1, 1)), 1.0, rtol=2e-7)      # the Rec. 601 weights sum to 1


def test_getnormal_is_the_normalised_raw_normal(oracle, default_setup):          # shader:80-82
    """getnormal(p) = normalize(getnormalraw(p)): on the sphere of the default scene (centre (-0.584, 0.049, 0.602), radius .5)
    the normal at a surface point is the radial direction, to the accuracy the 5e-5 differences allow in binary32"""
    _, _, props = default_setup(16, 9)
    l = oracle.lib()
    uni = oracle.Uniforms(props)
    c = np.array([-0.584, 0.049, 0.602])
    d = np.array([0.0, 0.0, -1.0])                                              # the side facing the camera, away from the box
    n, raw = f32([0, 0, 0]), f32([0, 0, 0])
    l.rmo_getnormal(uni.ref, fp(f32(c + 0.5 * d)), fp(n), fp(raw))
    np.testing.assert_allclose(np.linalg.norm(n), 1.0, rtol=1e-6)
    assert float(np.dot(n, d)) > 0.99
    # raw is 2e times the gradient (not divided, shader:63-67): |raw| ~ 1e-4
    assert 0.5e-4 < float(np.linalg.norm(raw)) < 2e-4
    np.testing.assert_allclose(n, raw / np.float32(np.sqrt(np.float32(np.dot(raw, raw)))), rtol=1e-6)


def test_composite_body_of_the_commented_line(oracle, default_setup):            # cginc:346-349
    """_RmSceneBody 6 = mandelbulb(rotate(p, (0,0,T)), menger(rotate(p, (T/4, 1.4T/4, 1.6T/4))).w + 1.1) on the world-space p.
    At T = 0 both rotations are the identity, so the value is mandelbulb(p, menger(p).w + 1.1) of the probes below."""
    _, _, props = default_setup(16, 9)
    l = oracle.lib()
    out, m, b = f32([0, 0, 0, 0]), f32([0, 0, 0, 0]), f32([0, 0, 0, 0])
    props = dict(props, _RmSceneBody=6.0, _Time=f32([0, 0, 0, 0]))
    uni = oracle.Uniforms(props)
    for p in ((0.3, -0.2, 0.9), (1.4, 0.1, -0.3), (0.05, -0.02, 2.5)):
        l.rmo_scene(uni.ref, p3(*p), fp(out))
        l.rmo_menger(uni.ref, p3(*p), fp(m))
        l.rmo_mandelbulb(p3(*p), float(np.float32(m[3]) + np.float32(1.1)), fp(b))
        assert out.tolist() == b.tolist(), p
    # T != 0: the bulb's argument is p turned about z by T (cginc:51-60 with a = (0, 0, T)); the sponge's by the three angles
    T = np.float32(0.8)
    uni_t = oracle.Uniforms(dict(props, _Time=f32([T / 20, T, 2 * T, 3 * T])))
    p = np.array([0.3, -0.2, 0.9], dtype=np.float32)
    l.rmo_scene(uni_t.ref, fp(p), fp(out))
    cz, sz = np.float32(math.cos(float(T))), np.float32(math.sin(float(T)))
    pz = f32([cz * p[0] - sz * p[1], sz * p[0] + cz * p[1], p[2]])
    ax, ay, az = T * np.float32(.25), T * np.float32(1.4) * np.float32(.25), T * np.float32(1.6) * np.float32(.25)
    uni_m = oracle.Uniforms(props)                                               # menger reads _FractalRotationX/Y/Z, unchanged by T

    def rot(v, a):
        cx, sx, cy, sy, cz_, sz_ = (np.float32(f(float(t))) for t in a for f in (math.cos, math.sin))
        a1 = f32([v[0], cx * v[1] - sx * v[2], sx * v[1] + cx * v[2]])
        b1 = f32([cy * a1[0] + sy * a1[2], a1[1], -sy * a1[0] + cy * a1[2]])
        return f32([cz_ * b1[0] - sz_ * b1[1], sz_ * b1[0] + cz_ * b1[1], b1[2]])
    l.rmo_menger(uni_m.ref, fp(rot(p, (ax, ay, az))), fp(m))
    l.rmo_mandelbulb(fp(pz), float(np.float32(m[3]) + np.float32(1.1)), fp(b))
    np.testing.assert_allclose(out, b, rtol=2e-6, atol=1e-7)
    # _RmAnimateFractal: the sponge's own fold angles become sin(T)*.5, sin(1.414 T)*.5, sin(1.618 T)*.5 (cginc:346-348)
    uni_a = oracle.Uniforms(dict(props, _Time=f32([T / 20, T, 2 * T, 3 * T]), _RmAnimateFractal=1.0))
    anim = dict(props, _FractalRotationX=float(np.float32(math.sin(float(T))) * np.float32(.5)),
                _FractalRotationY=float(np.float32(math.sin(float(T * np.float32(1.414)))) * np.float32(.5)),
                _FractalRotationZ=float(np.float32(math.sin(float(T * np.float32(1.618)))) * np.float32(.5)),
                _Time=f32([T / 20, T, 2 * T, 3 * T]))
    a1, a2 = f32([0, 0, 0, 0]), f32([0, 0, 0, 0])
    l.rmo_scene(uni_a.ref, fp(p), fp(a1))
    l.rmo_scene(oracle.Uniforms(anim).ref, fp(p), fp(a2))
    assert a1.tolist() == a2.tolist()
    assert a1.tolist() != out.tolist()
