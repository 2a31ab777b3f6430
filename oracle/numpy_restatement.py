"""Second, independently written restatement of the reference's march in NumPy
float32, vectorised over pixels.  It exists only to cross-check oracle/rm_oracle.c
(N-version check: the reference ships no golden image to pin either of them).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED BY THE REFERENCE.

NumPy's float32 ufuncs are IEEE-754 binary32 with one rounding per operation and
no fused multiply-add, i.e. the same arithmetic model as the C oracle.  Citations:
"cginc" = Assets/Raymarching.cginc, "shader" = Assets/Raymarching.shader.
"""
from __future__ import annotations

import numpy as np

f = np.float32


def row(m, i):
    """Row i of a column-major Unity Matrix4x4 (the shader's M[i])."""
    m = np.asarray(m, dtype=f).reshape(16)
    return m[i::4]


def dot(ax, ay, az, bx, by, bz):
    return ax * bx + ay * by + az * bz


def length3(x, y, z):
    return np.sqrt(dot(x, y, z, x, y, z))


def length2(x, y):
    return np.sqrt(x * x + y * y)


def lerp(a, b, t):
    return a + t * (b - a)


def clamp(x, lo, hi):
    return np.fmin(np.fmax(x, f(lo)), f(hi))


def rotate_q(vx, vy, vz, q):
    """cginc:62-66"""
    ux, uy, uz, qw = -q[0], -q[1], -q[2], q[3]
    a = f(2) * dot(ux, uy, uz, vx, vy, vz)
    c = qw * qw - dot(ux, uy, uz, ux, uy, uz)
    w2 = f(2) * qw
    cx = uy * vz - uz * vy
    cy = uz * vx - ux * vz
    cz = ux * vy - uy * vx
    return (ux * a + vx * c) + cx * w2, (uy * a + vy * c) + cy * w2, (uz * a + vz * c) + cz * w2


def sphere(x, y, z, r=1.0):                       # cginc:169-171
    return length3(x, y, z) - f(r) * f(.5)


def box(x, y, z, b=(1.0, 1.0, 1.0)):              # cginc:173-177
    bx, by, bz = f(b[0]) * f(.5), f(b[1]) * f(.5), f(b[2]) * f(.5)
    dx, dy, dz = np.abs(x) - bx, np.abs(y) - by, np.abs(z) - bz
    outside = length3(np.fmax(dx, f(0)), np.fmax(dy, f(0)), np.fmax(dz, f(0)))
    return outside + np.fmin(np.fmax(dx, np.fmax(dy, dz)), f(0))


def torus(x, y, z, tx=.5, ty=.25):                # cginc:179-182
    return length2(length2(x, y) - f(tx), z) - f(ty)


def link(x, y, z, le, r1, r2):                    # cginc:184-187
    qy = np.fmax(np.abs(y) - f(le), f(0))
    return length2(length2(x, qy) - f(r1), z) - f(r2)


def cylinder(x, y, z, h=.5, r=.5):                # cginc:189-192
    dx = np.abs(length2(x, z)) - f(h)
    dy = np.abs(y) - f(r)
    return np.fmin(np.fmax(dx, dy), f(0)) + length2(np.fmax(dx, f(0)), np.fmax(dy, f(0)))


def smin_w(a, b, k=.5):                           # cginc:110-113, distance lane
    k = f(k)
    h = clamp(f(.5) + f(.5) * (b - a) / k, 0, 1)
    return lerp(b, a, h) - k * h * (f(1) - h), h


def smax_w(a, b, k=.5):                           # cginc:120-123, distance lane
    k = f(k)
    h = clamp(f(.5) + f(.5) * (a - b) / k, 0, 1)
    return -lerp(-b, -a, h) - k * h * (f(1) - h), h


def blend(c1, c2, h):                             # cginc:101-103
    mx, my, mz = lerp(c1[0], c2[0], h), lerp(c1[1], c2[1], h), lerp(c1[2], c2[2], h)
    inv = f(1) / np.sqrt(dot(mx, my, mz, mx, my, mz))
    l = lerp(length3(*c1), length3(*c2), h)
    return (mx * inv) * l, (my * inv) * l, (mz * inv) * l


# ---- canonical transcendental functions: float -> float through IEEE double arithmetic in a fixed
# operation order (the definition is stated with the C oracle); float64 ufuncs round once per operation
d = np.float64
PIO2_HI, PIO2_LO = 1.57079632673412561417e+00, 6.07710050650619224932e-11
PI, PIO2, PIO4 = 3.14159265358979311600e+00, 1.57079632679489655800e+00, 7.85398163397448278999e-01
LN2, LN2_HI, LN2_LO = 6.93147180559945286227e-01, 6.93147180369123816490e-01, 1.90821492927058770002e-10
FACT = [1.0]
for _n in range(1, 18):
    FACT.append(FACT[-1] * _n)


def _sin_poly(r):
    q = r * r
    p = np.full_like(r, 1.0 / FACT[17])
    for n, sign in ((15, -1), (13, 1), (11, -1), (9, 1), (7, -1), (5, 1), (3, -1)):
        p = p * q + sign * (1.0 / FACT[n])
    return r + r * (q * p)


def _cos_poly(r):
    q = r * r
    p = np.full_like(r, 1.0 / FACT[16])
    for n, sign in ((14, -1), (12, 1), (10, -1), (8, 1), (6, -1), (4, 1)):
        p = p * q + sign * (1.0 / FACT[n])
    p = p * q - 0.5
    return 1.0 + q * p


def _reduce(x):
    k = np.floor(x * (1.0 / PIO2) + 0.5)
    r = (x - k * PIO2_HI) - k * PIO2_LO
    n = (k - 4.0 * np.floor(k * 0.25))
    with np.errstate(invalid="ignore"):
        n = np.where(np.isfinite(n), n, 0).astype(np.int64)
    return r, n


def c_sin(xf):
    r, n = _reduce(xf.astype(d))
    v = np.where(n & 1, _cos_poly(r), _sin_poly(r))
    return np.where(n & 2, -v, v).astype(f)


def c_cos(xf):
    r, n = _reduce(xf.astype(d))
    v = np.where(n & 1, _sin_poly(r), _cos_poly(r))
    return np.where((n + 1) & 2, -v, v).astype(f)


def _atan01(t):
    big = t > 0.41421356237309503
    u = np.where(big, (t - 1.0) / (t + 1.0), t)
    base = np.where(big, PIO4, 0.0)
    q = u * u
    p = np.full_like(t, 1.0 / 43.0)
    for n in range(41, 0, -2):
        p = 1.0 / float(n) - q * p
    return base + u * p


def _atan2d(y, x):
    ay, ax = np.abs(y), np.abs(x)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(ay > ax, PIO2 - _atan01(ax / ay), np.where(ax > 0.0, _atan01(ay / ax), ay / ax))
    a = np.where(x < 0.0, PI - a, a)
    return np.where(y < 0.0, -a, a)


def c_atan2(yf, xf):
    return _atan2d(yf.astype(d), xf.astype(d)).astype(f)


def _one_minus_sq(v):
    t = 1.0 - v * v
    return np.where(t > 0.0, t, 0.0)


def c_acos(vf):
    v = vf.astype(d)
    return _atan2d(np.sqrt(_one_minus_sq(v)), v).astype(f)


def c_asin(vf):
    v = vf.astype(d)
    return _atan2d(v, np.sqrt(_one_minus_sq(v))).astype(f)


def _logd(x):
    m, e = np.frexp(x)
    m = m * 2.0
    e = e - 1
    big = m > 1.4142135623730951
    m = np.where(big, m * 0.5, m)
    e = np.where(big, e + 1, e)
    s = (m - 1.0) / (m + 1.0)
    q = s * s
    p = np.full_like(x, 1.0 / 25.0)
    for n in range(23, 0, -2):
        p = 1.0 / float(n) + q * p
    return e.astype(d) * LN2 + 2.0 * s * p


def _expd(y):
    ok = (y < 700.0) & (y >= -700.0)
    yy = np.where(ok, y, 0.0)
    k = np.floor(yy * (1.0 / LN2) + 0.5)
    fr = (yy - k * LN2_HI) - k * LN2_LO
    p = np.full_like(y, 1.0 / FACT[14])
    for n in range(13, -1, -1):
        p = 1.0 / FACT[n] + fr * p
    out = np.ldexp(p, k.astype(np.int64))
    over = np.where(y > 0.0, np.inf, y)                   # overflow; NaN stays NaN
    return np.where(ok, out, np.where(y < -700.0, 0.0, over))


def c_log(xf):
    x = xf.astype(d)
    pos = (xf > 0) & np.isfinite(xf)
    with np.errstate(all="ignore"):
        v = _logd(np.where(pos, x, 1.0)).astype(f)
    special = np.where(xf == 0, f(-np.inf), np.where(np.isposinf(xf), f(np.inf), f(np.nan)))
    return np.where(pos, v, special).astype(f)


def c_pow(xf, e):
    """x >= 0 (a length), e a float32 scalar or an array like x"""
    e = np.asarray(e, dtype=f)
    pos = (xf > 0) & np.isfinite(xf)
    with np.errstate(all="ignore"):
        v = _expd(e.astype(d) * _logd(np.where(pos, xf.astype(d), 1.0))).astype(f)
    at0 = np.where(e > 0, f(0), np.where(e == 0, f(1), f(np.inf)))
    atinf = np.where(e > 0, f(np.inf), np.where(e == 0, f(1), f(0)))
    special = np.where(xf == 0, at0, np.where(np.isposinf(xf), atinf, f(np.nan)))
    return np.where(pos, v, special).astype(f)


def _bulb(x, y, z, e, brot):
    """cginc:194-227 (mandelbulb, brot False) and cginc:229-262 (mandelbrot, brot True); e a scalar or one exponent per point"""
    e_all = np.broadcast_to(np.asarray(e, dtype=f), x.shape)
    px, py, pz = x, y, z
    zx, zy, zz = x.copy(), y.copy(), z.copy()
    dr = np.ones_like(x)
    r = np.zeros_like(x)
    o = np.full_like(x, 10)
    o2, o3 = o.copy(), o.copy()
    live = np.ones(x.shape, dtype=bool)
    for _ in range(12):
        idx = np.nonzero(live)[0]
        if idx.size == 0:
            break
        ax, ay, az = zx[idx], zy[idx], zz[idx]
        rr = length3(ax, ay, az)
        r[idx] = rr
        o[idx] = np.fmin(o[idx], length3(ax, ay, az))
        if brot:
            o2[idx] = np.fmin(o2[idx], length3(ax - f(1), ay, az))
            o3[idx] = np.fmin(o3[idx], length3(ax - f(2), ay, az))
        else:
            o2[idx] = np.fmin(o2[idx], length3(ax, ay, az - f(1)))
            o3[idx] = np.fmin(o3[idx], length3(ax, ay, az - f(2)))
        go = ~(rr > f(10))
        live[idx[~go]] = False
        idx, ax, ay, az, rr = idx[go], ax[go], ay[go], az[go], rr[go]
        if idx.size == 0:
            break
        e = e_all[idx]
        with np.errstate(all="ignore"):
            ratio = az / rr
        if brot:
            theta, phi = c_atan2(ay, ax), c_asin(ratio)
        else:
            theta, phi = c_acos(ratio), c_atan2(ay, ax)
        dr[idx] = c_pow(rr, e - f(1)) * e * dr[idx] + f(1)
        zr = c_pow(rr, e)
        theta, phi = theta * e, phi * e
        if brot:
            nx, ny, nz = zr * (c_cos(theta) * c_cos(phi)), zr * (c_sin(theta) * c_cos(phi)), zr * (-c_sin(phi))
        else:
            nx, ny, nz = zr * (c_sin(theta) * c_cos(phi)), zr * (c_sin(phi) * c_sin(theta)), zr * c_cos(theta)
        zx[idx], zy[idx], zz[idx] = nx + px[idx], ny + py[idx], nz + pz[idx]
    with np.errstate(all="ignore"):
        w = f(.5) * c_log(r) * r / dr
    return (o, o2, o3), w


def euler_matrices(ax, ay, az):
    """cginc:53-57: the three float3x3 of rotate(float3 v, float3 a), zeros and ones included"""
    import math
    cx, sx = f(math.cos(float(ax))), f(math.sin(float(ax)))
    cy, sy = f(math.cos(float(ay))), f(math.sin(float(ay)))
    cz, sz = f(math.cos(float(az))), f(math.sin(float(az)))
    o, z = f(1), f(0)
    mx = ((o, z, z), (z, cx, -sx), (z, sx, cx))
    my = ((cy, z, sy), (z, o, z), (-sy, z, cy))
    mz = ((cz, -sz, z), (sz, cz, z), (z, z, o))
    return mx, my, mz


def mul33v(m, x, y, z):
    """mul(float3x3, float3): one dot product per row, every term kept"""
    return tuple(m[r][0] * x + m[r][1] * y + m[r][2] * z for r in range(3))


def rotate_euler(x, y, z, ms):                    # cginc:51-60: mul(mz, mul(my, mul(mx, v)))
    mx, my, mz = ms
    return mul33v(mz, *mul33v(my, *mul33v(mx, x, y, z)))


def menger(x, y, z, ms):                          # cginc:295-313
    import math
    o = np.full_like(x, 50)
    o2, o3 = o.copy(), o.copy()
    for _ in range(12):
        x, y, z = np.abs(x), np.abs(y), np.abs(z)
        s = x < y
        x, y = np.where(s, y, x), np.where(s, x, y)
        s = y < z
        y, z = np.where(s, z, y), np.where(s, y, z)
        s = x < y
        x, y = np.where(s, y, x), np.where(s, x, y)
        x, y, z = rotate_euler(x, y, z, ms)
        x, y, z = x * f(3) - f(2), y * f(3) - f(2), z * f(3) - f(2)
        z = np.where(z < f(-1), z + f(2), z)
        o = np.fmin(o, length3(x, y, z))
        o2 = np.fmin(o2, length3(x, y - f(.5), z))
        o3 = np.fmin(o3, length3(x, y, z - f(.5)))
    return (o, o2, o3), (np.sqrt(x * x + y * y + z * z) - f(1.5)) * f(math.pow(3.0, -12.0))


def sierpinski(x, y, z, scale, ms):               # cginc:264-293
    import math
    scale = f(scale)
    x, y, z = x.copy(), y.copy(), z.copy()
    r = x * x + y * y + z * z
    bail = f(20)
    o = np.full_like(x, bail)
    o2, o3 = o.copy(), o.copy()
    count = np.zeros(x.shape, dtype=np.int64)
    for _ in range(15):
        live = r < bail
        if not live.any():
            break
        lx, ly, lz = x[live], y[live], z[live]
        c = lx + ly < 0
        lx, ly = np.where(c, -ly, lx), np.where(c, -lx, ly)
        c = lx + lz < 0
        lx, lz = np.where(c, -lz, lx), np.where(c, -lx, lz)
        c = ly + lz < 0
        ly, lz = np.where(c, -lz, ly), np.where(c, -ly, lz)
        lx, ly, lz = rotate_euler(lx, ly, lz, ms)
        lx = scale * lx - (scale - f(1))
        ly = scale * ly - (scale - f(1))
        lz = scale * lz - (scale - f(1))
        x[live], y[live], z[live] = lx, ly, lz
        r[live] = lx * lx + ly * ly + lz * lz
        o[live] = np.fmin(o[live], length3(lx - f(1), ly, lz))
        o2[live] = np.fmin(o2[live], length3(lx, ly - f(1), lz))
        o3[live] = np.fmin(o3[live], length3(lx, ly, lz - f(1)))
        count[live] += 1
    pows = np.array([math.pow(float(scale), -float(i)) for i in range(16)]).astype(f)
    return (o, o2, o3), (np.sqrt(r) - f(2)) * pows[count]


def frac(x):
    return x - np.floor(x)


def rand2(x, y):                                  # cginc:125-127
    return frac(c_sin(x * f(12.9898) + y * f(78.233)) * f(43758.5453))


def noise2(x, y):                                 # cginc:129-149
    ix, iy = np.floor(x), np.floor(y)
    fx, fy = frac(x), frac(y)
    a, b = rand2(ix, iy), rand2(ix + f(1), iy + f(0))
    c, dd = rand2(ix + f(0), iy + f(1)), rand2(ix + f(1), iy + f(1))
    ux, uy = fx * fx * (f(3) - f(2) * fx), fy * fy * (f(3) - f(2) * fy)
    return lerp(a, b, ux) + (c - a) * uy * (f(1) - ux) + (dd - b) * ux * uy


def terrain_w(x, y):                              # cginc:315-330
    px, py = x * f(0.003) / f(250), y * f(0.003) / f(250)
    a, b = np.zeros_like(x), f(1)
    dx, dy = np.zeros_like(x), np.zeros_like(x)
    for _ in range(3):
        n = noise2(px, py)
        dx, dy = dx + n, dy + f(0)
        a = a + b * n / (f(1) + (dx * dx + dy * dy))
        b = b * f(.5)
        px, py = f(.8) * px * f(2), f(-.6) * py * f(2)
    return f(250) * f(120) * a


class Scene:
    COLORS = ((1, .25, .25), (.25, 1, .25), (.25, .25, 1), (1, 1, .25))     # cginc:356-359

    def __init__(self, props, prims=None):
        if prims:
            self.prims = [dict(p) for p in prims]
        else:
            self.prims = []
            par = [(1.0,), (1.0, 1.0, 1.0), (5.0, 5.0, .1), (.5, .5)]
            for i in range(4):
                self.prims.append({"type": i, "pos": row(props["_Positions"], i)[:3],
                                   "quat": row(props["_Rotations"], i), "scale": row(props["_Scales"], i)[:3],
                                   "color": self.COLORS[i], "params": par[i]})
        for p in self.prims:
            for k in ("pos", "quat", "scale", "color", "params"):
                p[k] = np.asarray(p[k], dtype=f)
        # fractal bodies: wrapped like a primitive, with slot 0 of the three matrices
        self.body = int(props.get("_RmSceneBody", 0))
        self.slot0 = {"pos": row(props["_Positions"], 0)[:3], "quat": row(props["_Rotations"], 0),
                      "scale": row(props["_Scales"], 0)[:3]}
        self.fscale = props.get("_FractalScale", 2.0)
        self.fpower = props.get("_RmFractalPower", 0.0)
        if not self.fpower > 0:
            self.fpower = 7.0 if self.body == 3 else 2.0
        import math
        t = f(np.asarray(props.get("_Time", (0, 0, 0, 0)), dtype=f)[1])
        angles = [f(props.get("_FractalRotationX", 0.0)), f(props.get("_FractalRotationY", 0.0)), f(props.get("_FractalRotationZ", 0.0))]
        if props.get("_RmAnimateFractal", 0.0):                 # cginc:346-348
            angles = [f(math.sin(float(t))) * f(.5), f(math.sin(float(t * f(1.414)))) * f(.5), f(math.sin(float(t * f(1.618)))) * f(.5)]
        self.fms = euler_matrices(*angles)
        # cginc:349: rotate(p, float3(0, 0, _Time.y)) and rotate(p, float3(_Time.y*.25, _Time.y*1.4*.25, _Time.y*1.6*.25))
        self.ms_time = euler_matrices(f(0), f(0), t)
        self.ms_sponge = euler_matrices(t * f(.25), t * f(1.4) * f(.25), t * f(1.6) * f(.25))
        self.dom_op = int(props.get("_RmDomainOp", 0))
        self.dom_period = [f(v) for v in list(props.get("_RmDomainPeriod", (0, 0, 0, 0)))[:3]]
        self.dom_size = props.get("_RmDomainSize", 1.0)

    def domain(self, x, y, z):
        """the domain operator in front of scene(): repetition (cginc:84-99), boxfold (:156-159),
        spherefold (:161-165)"""
        if self.dom_op == 1:
            def shmod(v, m):
                m = f(m)
                hm = m * f(.5)
                r = np.fmod(v + hm, m)                      # exact remainder with the sign of the dividend
                return np.where(r < 0, r + m, r) - hm
            x, y, z = (shmod(v, m) if m > 0 else v for v, m in zip((x, y, z), self.dom_period))
        elif self.dom_op == 2:
            b = f(self.dom_size) * f(.5)
            x, y, z = (clamp(v, -b, b) * f(2) - v for v in (x, y, z))
        elif self.dom_op == 3:
            R = f(self.dom_size)
            r = length3(x, y, z)
            rr = r * r
            with np.errstate(all="ignore"):
                x, y, z = (np.where(r < R, v * R * R / rr, v) for v in (x, y, z))
        return x, y, z

    def prim(self, p, x, y, z):
        """cginc:336-354: rotate(p - pos, q) / scale -> primitive -> * min(scale)"""
        lx, ly, lz = rotate_q(x - p["pos"][0], y - p["pos"][1], z - p["pos"][2], p["quat"])
        s = p["scale"]
        lx, ly, lz = lx / s[0], ly / s[1], lz / s[2]
        t, a = int(p["type"]), p["params"]
        if t == 0:
            d = sphere(lx, ly, lz, a[0])
        elif t == 1:
            d = box(lx, ly, lz, a[:3])
        elif t == 2:
            d = link(lx, ly, lz, a[0], a[1], a[2])
        elif t == 3:
            d = cylinder(lx, ly, lz, a[0], a[1])
        else:
            d = torus(lx, ly, lz, a[0], a[1])
        d = d * np.fmin(s[0], np.fmin(s[1], s[2]))
        return -d if int(p.get("op", 0)) & 2 else d            # NEGATE bit: complement solid

    def eval(self, x, y, z, want_rgb=False):
        """cginc:334-364 generalised: the pairwise tree in index order with an odd element carried
        up is the left-complete tree -- a node over n leaves has the largest power of two below n
        leaves on its left.  A node is smax (cginc:120-123) when the first leaf of its right
        operand carries the SMAX bit, smin (cginc:110-113) otherwise."""
        if self.dom_op:
            x, y, z = self.domain(x, y, z)
        if self.body == 6:                        # cginc:349 on the world-space p, no wrapper
            _, mw = menger(*rotate_euler(x, y, z, self.ms_sponge), self.fms)
            return _bulb(*rotate_euler(x, y, z, self.ms_time), mw + f(1.1), False)
        if self.body:
            p0 = self.slot0
            lx, ly, lz = rotate_q(x - p0["pos"][0], y - p0["pos"][1], z - p0["pos"][2], p0["quat"])
            s = p0["scale"]
            lx, ly, lz = lx / s[0], ly / s[1], lz / s[2]
            if self.body == 1:
                c, dist = menger(lx, ly, lz, self.fms)
            elif self.body == 2:
                c, dist = sierpinski(lx, ly, lz, self.fscale, self.fms)
            elif self.body == 5:                  # t = terrain(p.xz); float4(t.rgb, p.y - t.w)
                one = np.ones_like(lx)
                c, dist = (one, one, one), ly - terrain_w(lx, lz)
            else:
                c, dist = _bulb(lx, ly, lz, self.fpower, self.body == 4)
            return c, dist * np.fmin(s[0], np.fmin(s[1], s[2]))

        def node(lo, hi):
            if hi - lo == 1:
                p = self.prims[lo]
                d = self.prim(p, x, y, z)
                c = tuple(np.full_like(d, p["color"][j]) for j in range(3)) if want_rgb else None
                return c, d
            split = 1
            while split * 2 < hi - lo:
                split *= 2
            (ca, a), (cb, b) = node(lo, lo + split), node(lo + split, hi)
            if int(self.prims[lo + split].get("op", 0)) & 1:
                w, h = smax_w(a, b)
                c = tuple(lerp(cb[j], ca[j], h) for j in range(3)) if want_rgb else None
            else:
                w, h = smin_w(a, b)
                c = blend(cb, ca, h) if want_rgb else None
            return c, w
        return node(0, len(self.prims))


def march(scene, ox, oy, oz, dx, dy, dz, depth, steps, sf, shadow=False):
    """cginc:367-392 (raymarch) and the loop of shader:102-123 (getshadow, k=16)
    over arrays of rays; returns hit, evals, length, res."""
    n = ox.shape[0]
    dm = np.zeros(n, dtype=f)
    res = np.ones(n, dtype=f)
    hit = np.zeros(n, dtype=bool)
    evals = np.zeros(n, dtype=np.int32)
    left_by_break = np.zeros(n, dtype=bool)
    active = np.arange(n)
    i = 0
    with np.errstate(all="ignore"):
        while active.size and f(i) < steps:
            a = active
            t = dm[a]
            _, dts = scene.eval(ox[a] + dx[a] * t, oy[a] + dy[a] * t, oz[a] + dz[a] * t)
            evals[a] += 1
            if shadow:
                res[a] = np.fmin(res[a], f(16) * dts / t)
            t = t + np.abs(dts) * sf
            dm[a] = t
            contact = t * f(.0025)
            h = np.abs(dts) < contact
            hit[a[h]] = True
            far = t > depth[a]
            left_by_break[a[h | far]] = True
            active = a[~h & ~far]
            i += 1
    march.last_steps = evals - left_by_break          # r.steps: the loop counter after the loop (cginc:387)
    return hit, evals, dm, res


def unorm8(x):
    with np.errstate(invalid="ignore"):
        x = np.fmin(np.fmax(x, f(0)), f(1))
    return np.floor(x * f(255) + f(.5)).astype(np.uint8)


def pixel_rays(props, w, h):
    """vert (cginc:26-33) + rasteriser interpolation to pixel centres, uv origin bottom-left."""
    fc, c2w = props["_FrustrumCorners"], props["_CameraInvViewMatrix"]
    corners = []
    for i in range(4):
        v = row(fc, i)
        corners.append([dot(row(c2w, j)[0], row(c2w, j)[1], row(c2w, j)[2], v[0], v[1], v[2]) for j in range(3)])
    tl, tr, br, bl = corners
    xs = (np.arange(w, dtype=f) + f(.5)) / f(w)
    ys = (np.arange(h, dtype=f) + f(.5)) / f(h)
    uu, vv = np.meshgrid(xs, ys)
    uu, vv = uu.reshape(-1), vv.reshape(-1)
    rd = [lerp(lerp(bl[j], br[j], uu), lerp(tl[j], tr[j], uu), vv) for j in range(3)]
    return rd


def shadow_lookup(props, shadow_map, qx, qy, qz):
    """What reads _ShadowMap / _ShadowMapVPMatrix (declared at shader:45-46, never read by the reference): 1 lit, 0 shadowed.
    Mode 1: c = mul(VP, float4(q,1)), uv = c.xy/c.w*.5+.5, eye depth c.w against the stored r.length; mode 2: coordinates
    along the edges of the ShadowMapDirectional rectangle and along its forward."""
    mode = int(props["_RmShadowMode"])
    sm = np.asarray(shadow_map, dtype=f)
    sh, sw = sm.shape
    with np.errstate(all="ignore"):
        if mode == 1:
            vp = props["_ShadowMapVPMatrix"]
            r0, r1, r3 = row(vp, 0), row(vp, 1), row(vp, 3)
            cx = ((r0[0] * qx + r0[1] * qy) + r0[2] * qz) + r0[3]
            cy = ((r1[0] * qx + r1[1] * qy) + r1[2] * qz) + r1[3]
            cw = ((r3[0] * qx + r3[1] * qy) + r3[2] * qz) + r3[3]
            front = cw > 0
            uu = (cx / cw) * f(.5) + f(.5)
            vv = (cy / cw) * f(.5) + f(.5)
            d = cw
        else:
            k = props["_RmShadowMapCorners"]
            tl, br, bl = row(k, 0), row(k, 2), row(k, 3)
            e1 = [br[j] - bl[j] for j in range(3)]
            e2 = [tl[j] - bl[j] for j in range(3)]
            fw = np.asarray(props["_RmShadowMapForward"], dtype=f)
            rx, ry, rz = qx - bl[0], qy - bl[1], qz - bl[2]
            uu = dot(rx, ry, rz, e1[0], e1[1], e1[2]) / dot(e1[0], e1[1], e1[2], e1[0], e1[1], e1[2])
            vv = dot(rx, ry, rz, e2[0], e2[1], e2[2]) / dot(e2[0], e2[1], e2[2], e2[0], e2[1], e2[2])
            d = dot(rx, ry, rz, fw[0], fw[1], fw[2])
            front = np.ones(qx.shape, dtype=bool)
        inside = front & (uu >= 0) & (uu < 1) & (vv >= 0) & (vv < 1)
        ix = np.where(inside, np.floor(uu * f(sw)), 0).astype(np.int64)
        iy = np.where(inside, np.floor(vv * f(sh)), 0).astype(np.int64)
        ix, iy = np.minimum(ix, sw - 1), np.minimum(iy, sh - 1)
        stored = sm[iy, ix]
        dark = inside & (d > stored * (f(1) + f(props["_RmShadowBias"])))
    return np.where(dark, f(0), f(1)).astype(f)


def depth_pass(props, w, h, prims=None, corners=None, forward=None):
    """The depth passes: raymarch() (cginc:367-392) along the rays of the bound camera (ShadowMap.Render, ShadowMap.cs:50-56)
    or, with corners/forward, along `forward` from the rectangle of ShadowMapDirectional.cs:25-34; r.length is written."""
    scene = Scene(props, prims)
    n = w * h
    if corners is None:
        rdx, rdy, rdz = pixel_rays(props, w, h)
        cam = np.asarray(props["_WorldSpaceCameraPos"], dtype=f)
        ox, oy, oz = (np.full(n, cam[j], dtype=f) for j in range(3))
    else:
        c = np.asarray(corners, dtype=f).reshape(4, 3)                  # TL, TR, BR, BL
        xs = (np.arange(w, dtype=f) + f(.5)) / f(w)
        ys = (np.arange(h, dtype=f) + f(.5)) / f(h)
        uu, vv = np.meshgrid(xs, ys)
        uu, vv = uu.reshape(-1), vv.reshape(-1)
        ox, oy, oz = [lerp(lerp(c[3][j], c[2][j], uu), lerp(c[0][j], c[1][j], uu), vv) for j in range(3)]
        fw = np.asarray(forward, dtype=f)
        rdx, rdy, rdz = (np.full(n, fw[j], dtype=f) for j in range(3))
    steps = np.fmin(f(2000), f(props["_MaxSteps"]))
    far = f(np.asarray(props["_ProjectionParams"], dtype=f)[2])
    depth = np.fmin(np.full(n, far, dtype=f), f(props["_MaxDist"]))
    hit, _, length, _ = march(scene, ox, oy, oz, rdx, rdy, rdz, depth, steps, f(props["_StepFactor"]))
    return {"depth": length.reshape(h, w), "hit": hit.astype(np.uint8).reshape(h, w)}


def render(props, w, h, prims=None, src=None, scene_depth=None, shadow_map=None):
    """shader:149-216 (frag) for every pixel.  Returns rgba u8 [h,w,4], depth, hit, stats."""
    scene = Scene(props, prims)
    n = w * h
    rdx, rdy, rdz = pixel_rays(props, w, h)
    cam = np.asarray(props["_WorldSpaceCameraPos"], dtype=f)
    L = np.asarray(props["_WorldSpaceLightPos0"], dtype=f)
    ox, oy, oz = (np.full(n, cam[j], dtype=f) for j in range(3))
    steps = np.fmin(f(2000), f(props["_MaxSteps"]))
    sf = f(props["_StepFactor"])
    far = f(np.asarray(props["_ProjectionParams"], dtype=f)[2])
    depth = np.full(n, far, dtype=f) if scene_depth is None else np.asarray(scene_depth, dtype=f).reshape(-1)
    depth = np.fmin(depth, f(props["_MaxDist"]))
    hit, ev_p, length, _ = march(scene, ox, oy, oz, rdx, rdy, rdz, depth, steps, sf)

    if src is None:
        bg = np.asarray(props["_BackgroundColor"], dtype=f)
        col = [np.full(n, bg[j], dtype=f) for j in range(3)]
    else:
        s = np.asarray(src, dtype=np.uint8).reshape(n, 4)
        col = [s[:, j].astype(f) / f(255) for j in range(3)]

    # lines the reference keeps commented out in frag, switched on by properties (all off = the shader as is)
    fx_heat, fx_view = props.get("_RmStepHeat", 0) != 0, int(props.get("_RmDebugView", 0))
    fx_vol, fx_bounce = props.get("_RmVolumetricLight", 0) != 0, int(props.get("_RmBounce", 0))
    light_model, fx_scatter = int(props.get("_RmLightModel", 0)), props.get("_RmScatter", 0) != 0
    lp = np.asarray(props.get("_LightPosition", (0, 0, 0, 0)), dtype=f)
    r_steps = march.last_steps.astype(f)
    r_len, r_hit = length.copy(), hit.copy()          # the ray `r` that frag ends up with (the bounced one, if any)
    rox, roy, roz, rrx, rry, rrz = ox.copy(), oy.copy(), oz.copy(), rdx.copy(), rdy.copy(), rdz.copy()
    e = np.fmax(f(0), f(props["_NormalSampleScale"]))                                           # shader:51
    sd_full = np.fmin(f(2000), f(props["_MaxDist"]))

    def normal_raw(hx, hy, hz):                                                                 # shader:63-67
        z0 = np.zeros_like(hx)
        return (scene.eval(hx + e, hy + z0, hz + z0)[1] - scene.eval(hx - e, hy - z0, hz - z0)[1],
                scene.eval(hx + z0, hy + e, hz + z0)[1] - scene.eval(hx - z0, hy - e, hz - z0)[1],
                scene.eval(hx + z0, hy + z0, hz + e)[1] - scene.eval(hx - z0, hy - z0, hz - e)[1])

    idx = np.nonzero(hit)[0]
    ev_s = np.zeros(n, dtype=np.int32)
    if idx.size:
        t = length[idx]
        hx, hy, hz = ox[idx] + rdx[idx] * t, oy[idx] + rdy[idx] * t, oz[idx] + rdz[idx] * t     # shader:174
        nx, ny, nz = normal_raw(hx, hy, hz)
        with np.errstate(all="ignore"):
            inv = f(1) / np.sqrt(dot(nx, ny, nz, nx, ny, nz))
            ux, uy, uz = nx * inv, ny * inv, nz * inv
        contact = t * f(.0025)                       # value raymarch() left in _ContactThreshold (cginc:376)
        if fx_bounce:                                                                           # shader:178-187
            with np.errstate(all="ignore"):
                bx, by, bz = hx - (ux * contact) * f(2), hy - (uy * contact) * f(2), hz - (uz * contact) * f(2)
                ix, iy, iz = rdx[idx], rdy[idx], rdz[idx]
                # reflect(i, n) = i - (2*dot(i,n))*n
                tr = f(2) * dot(ix, iy, iz, ux, uy, uz)
                fx_, fy_, fz_ = ix - ux * tr, iy - uy * tr, iz - uz * tr
                # refract(i, n, eta): k = 1 - eta*eta*(1 - dot(n,i)^2); k < 0 -> 0
                eta = f(1.1)
                dn = dot(ux, uy, uz, ix, iy, iz)
                k = f(1) - (eta * eta) * (f(1) - dn * dn)
                tt = eta * dn + np.sqrt(np.where(k < 0, f(0), k))
                gx = np.where(k < 0, f(0), ix * eta - ux * tt)
                gy = np.where(k < 0, f(0), iy * eta - uy * tt)
                gz = np.where(k < 0, f(0), iz * eta - uz * tt)
                if fx_bounce == 2:                                                              # refract2, shader:142-147
                    e2 = f(2) - f(props["_RefractiveIndex"])
                    cosi = dot(ux, uy, uz, ix, iy, iz)
                    t2 = -cosi + e2 * cosi
                    gx, gy, gz = ix * e2 - ux * t2, iy * e2 - uy * t2, iz * e2 - uz * t2
                inside = scene.eval(hx, hy, hz)[1] >= 0
                nrx, nry, nrz = np.where(inside, gx, fx_), np.where(inside, gy, fy_), np.where(inside, gz, fz_)
            bh, _, bl, _ = march(scene, bx, by, bz, nrx, nry, nrz, np.full(idx.size, sd_full, dtype=f), steps, sf)
            r_steps[idx] = march.last_steps.astype(f)
            r_len[idx], r_hit[idx] = bl, bh
            rox[idx], roy[idx], roz[idx], rrx[idx], rry[idx], rrz[idx] = bx, by, bz, nrx, nry, nrz
            with np.errstate(all="ignore"):
                hx, hy, hz = bx + nrx * bl, by + nry * bl, bz + nrz * bl
                nx, ny, nz = normal_raw(hx, hy, hz)
                inv = f(1) / np.sqrt(dot(nx, ny, nz, nx, ny, nz))
                ux, uy, uz = nx * inv, ny * inv, nz * inv
                contact = bl * f(.0025)
        srgb, _ = scene.eval(hx, hy, hz, want_rgb=True)                                         # shader:193
        with np.errstate(all="ignore"):
            sx, sy, sz = hx + (ux * contact) * f(2), hy + (uy * contact) * f(2), hz + (uz * contact) * f(2)
        sd = np.full(idx.size, sd_full, dtype=f)
        lx, ly, lz = (np.full(idx.size, L[j], dtype=f) for j in range(3))
        if int(props.get("_RmShadowMode", 0)) != 0:
            res, evs = shadow_lookup(props, shadow_map, sx, sy, sz), 0
        else:
            _, evs, _, res = march(scene, sx, sy, sz, lx, ly, lz, sd, steps, sf, shadow=True)
        ev_s[idx] = evs
        with np.errstate(all="ignore"):
            shadow = (res - f(0)) / (f(1) - f(0)) * (f(1) - f(.25)) + f(.25)                    # cginc:80-82
            ndl = dot(ux, uy, uz, L[0], L[1], L[2]) * f(.5) + f(.5)
            if light_model:                                                                     # shader:84-100, light of :157-162
                qx, qy, qz = hx - lp[0], hy - lp[1], hz - lp[2]
                dist = np.sqrt(dot(qx, qy, qz, qx, qy, qz))
                dn = dot(ux, uy, uz, lp[0] - hx, lp[1] - hy, lp[2] - hz)
                if light_model == 2:
                    br = lerp(clamp(dn, 0, 1) * f(1), f(0), dist / f(1000))
                else:
                    dd = clamp(dn, -1, 1) * f(.5) + f(.5)
                    dd = dd * dd
                    br = lerp(dd * f(1), f(0), dist / f(1000))
                ndl = np.where(dist >= f(1000), f(0), br)
            nss = f(props["_NormalSampleScale"])
            ao = clamp(length3(nx / nss, ny / nss, nz / nss), 0, 1)                             # shader:125-127
            shaded = [((srgb[j] * ndl) * shadow) * ao for j in range(3)]                        # shader:199
            if fx_view == 1:
                shaded = [shadow, shadow, shadow]                                               # shader:200
            elif fx_view == 2:
                shaded = [ux * f(.5) + f(.5), uy * f(.5) + f(.5), uz * f(.5) + f(.5)]           # shader:201
            elif fx_view == 3:
                v = clamp(srgb[0] * f(2.5) - f(1.75), 0, 1)                                     # shader:198
                shaded = [v, v, v]
            keep = r_hit[idx]
            for j in range(3):
                col[j][idx[keep]] = shaded[j][keep]
    with np.errstate(all="ignore"):
        if fx_heat:                                                                             # shader:203
            hterm = r_steps / f(100) * sf
            col = [c + hterm for c in col]
        if fx_scatter:                                                                          # shader:129-140
            qx, qy, qz = rox - lp[0], roy - lp[1], roz - lp[2]
            bb = dot(rrx, rry, rrz, qx, qy, qz)
            cc = dot(qx, qy, qz, qx, qy, qz)
            ss = f(1) / np.sqrt(cc - bb * bb)
            one = np.ones_like(ss)
            sct = ss * (c_atan2((r_len + bb) * ss, one) - c_atan2(bb * ss, one))
            col = [c + sct for c in col]
        if fx_vol:                                                                              # shader:206-212
            la = np.zeros(n, dtype=f)
            lim = np.fmin(f(5), r_len)
            i = f(0)
            while True:
                act = np.nonzero(i * f(.1) < lim)[0]
                if act.size == 0:
                    break
                px, py, pz = rox[act] + (rrx[act] * i) * f(.1), roy[act] + (rry[act] * i) * f(.1), roz[act] + (rrz[act] * i) * f(.1)
                lx, ly, lz = (np.full(act.size, L[j], dtype=f) for j in range(3))
                _, _, _, res = march(scene, px, py, pz, lx, ly, lz, np.full(act.size, sd_full, dtype=f), steps, sf, shadow=True)
                la[act] = la[act] + res * f(.05)
                i = i + f(1)
            col = [c + la for c in col]
    rgba = np.empty((n, 4), dtype=np.uint8)
    for j in range(3):
        rgba[:, j] = unorm8(col[j])
    rgba[:, 3] = 255
    stats = {"n_pixels": n, "n_hit": int(hit.sum()), "sum_steps_p": int(ev_p.sum()), "sum_steps_s": int(ev_s.sum())}
    return {"rgba": rgba.reshape(h, w, 4), "depth": length.reshape(h, w), "hit": hit.astype(np.uint8).reshape(h, w),
            "stats": stats}
