/*
 * rm_oracle.c -- CPU restatement of the reference's fragment-shader march.
 *
 * TEST INFRASTRUCTURE ONLY (see rm_oracle.h).  PARITY UNPINNED BY THE REFERENCE.
 *
 * Each function cites the reference lines it restates, relative to the
 * reference checkout (Assets/Raymarching.cginc = "cginc", Assets/Raymarching.shader
 * = "shader", the C# scripts by file name).  Build: see oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#define _GNU_SOURCE
#include "rm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

/* ---- canonical intrinsic expansions ------------------------------------ */
static inline f3 v3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 div3(f3 a, f3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline f3 scl3(f3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline f3 neg3(f3 a) { return v3(-a.x, -a.y, -a.z); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }
static inline float len3(f3 a) { return sqrtf(dot3(a, a)); }
static inline float len2(float x, float y) { return sqrtf(dot2(x, y, x, y)); }
static inline f3 cross3(f3 a, f3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline f3 normalize3(f3 a) { float r = 1.0f / sqrtf(dot3(a, a)); return scl3(a, r); }
static inline float lerp1(float a, float b, float t) { return a + t * (b - a); }
static inline f3 lerp3(f3 a, f3 b, float t) {
    return v3(lerp1(a.x, b.x, t), lerp1(a.y, b.y, t), lerp1(a.z, b.z, t));
}
static inline float clamp1(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* row i of a column-major Unity Matrix4x4 (the shader's M[i], cginc:30,336-354) */
static inline f4 mrow(const float *m, int i) { f4 r = {m[i], m[i + 4], m[i + 8], m[i + 12]}; return r; }
static inline f3 xyz(f4 a) { return v3(a.x, a.y, a.z); }

/* ---- cginc:62-66  rotate(float3 v, float4 q) ---------------------------- */
static inline f3 rotate_q(f3 v, f4 q) {
    f3 u = neg3(xyz(q));
    f3 t1 = scl3(u, 2.0f * dot3(u, v));
    f3 t2 = scl3(v, q.w * q.w - dot3(u, u));
    f3 t3 = scl3(cross3(u, v), 2.0f * q.w);
    return add3(add3(t1, t2), t3);
}

/* ---- cginc:76-78  min(float3) ------------------------------------------- */
static inline float min3(f3 a) { return fminf(a.x, fminf(a.y, a.z)); }

/* ---- cginc:80-82  remap -------------------------------------------------- */
static inline float remap(float x, float o1, float o2, float n1, float n2) {
    return (x - o1) / (o2 - o1) * (n2 - n1) + n1;
}

/* ---- cginc:169-171  sphere ----------------------------------------------- */
static inline float sd_sphere(f3 p, float r) { return len3(p) - r * .5f; }

/* ---- cginc:173-177  box -------------------------------------------------- */
static inline float sd_box(f3 p, f3 b) {
    b = scl3(b, .5f);
    f3 d = v3(fabsf(p.x) - b.x, fabsf(p.y) - b.y, fabsf(p.z) - b.z);
    f3 dm = v3(fmaxf(d.x, 0.0f), fmaxf(d.y, 0.0f), fmaxf(d.z, 0.0f));
    return len3(dm) + fminf(fmaxf(d.x, fmaxf(d.y, d.z)), 0.0f);
}

/* ---- cginc:179-182  torus ------------------------------------------------ */
static inline float sd_torus(f3 p, float tx, float ty) {
    float qx = len2(p.x, p.y) - tx;
    return len2(qx, p.z) - ty;
}

/* ---- cginc:184-187  link (scalar result broadcast to float4) ------------- */
static inline float sd_link(f3 p, float le, float r1, float r2) {
    f3 q = v3(p.x, fmaxf(fabsf(p.y) - le, 0.0f), p.z);
    return len2(len2(q.x, q.y) - r1, q.z) - r2;
}

/* ---- cginc:189-192  cylinder (h acts as radius, r as half height) -------- */
static inline float sd_cylinder(f3 p, float h, float r) {
    float dx = fabsf(len2(p.x, p.z)) - h;
    float dy = fabsf(p.y) - r;
    return fminf(fmaxf(dx, dy), 0.0f) + len2(fmaxf(dx, 0.0f), fmaxf(dy, 0.0f));
}

/* ---- cginc:101-103  blend ------------------------------------------------ */
static inline f3 blend3(f3 c1, f3 c2, float k) {
    return scl3(normalize3(lerp3(c1, c2, k)), lerp1(len3(c1), len3(c2), k));
}

/* ---- cginc:110-113  smin(float4,float4,k) --------------------------------- */
static inline f4 smin4(f4 a, f4 b, float k) {
    float h = clamp1(0.5f + 0.5f * (b.w - a.w) / k, 0.0f, 1.0f);
    f3 c = blend3(xyz(b), xyz(a), h);
    f4 r = {c.x, c.y, c.z, lerp1(b.w, a.w, h) - k * h * (1.0f - h)};
    return r;
}

/* ---- cginc:120-123  smax(float4,float4,k): plain lerp of rgb, and the reference's
 * sign of the k*h*(1-h) term (subtracted, as in smin) ----------------------- */
static inline f4 smax4(f4 a, f4 b, float k) {
    float h = clamp1(0.5f + 0.5f * (a.w - b.w) / k, 0.0f, 1.0f);
    f3 c = lerp3(xyz(b), xyz(a), h);
    f4 r = {c.x, c.y, c.z, -lerp1(-b.w, -a.w, h) - k * h * (1.0f - h)};
    return r;
}

/* ---- helpers the reference defines and no live code calls (restated so that nothing of the .cginc is left without a
 * checked restatement) ---------------------------------------------------------------------------------------------
 * cginc:105-108  smin(float a, float b, k): the scalar overload, same arithmetic as the .w lane of the float4 one */
static inline float smin1(float a, float b, float k) {
    float h = clamp1(0.5f + 0.5f * (b - a) / k, 0.0f, 1.0f);
    return lerp1(b, a, h) - k * h * (1.0f - h);
}
/* cginc:115-118  smax(float a, float b, k).  Unlike the float4 overload (cginc:120-123, "-lerp(-b.w, -a.w, h)") the scalar
 * one returns lerp(-b, -a, h) WITHOUT the outer negation: for a >= b + k it returns -a, not a.  Restated as written. */
static inline float smax1(float a, float b, float k) {
    float h = clamp1(0.5f + 0.5f * (a - b) / k, 0.0f, 1.0f);
    return lerp1(-b, -a, h) - k * h * (1.0f - h);
}
/* cginc:68-74  sq */
static inline float sq1(float x) { return x * x; }
/* cginc:151-154  luminance: the Rec. 601 weights of the live line */
static inline float luminance3(f3 c) { return c.x * 0.2990f + c.y * 0.5870f + c.z * 0.1140f; }

/* primitive wrapper: p_i = rotate(p - pos, q);  o = prim(p_i / scale) * min(scale)
 * (cginc:336-354) */
static inline float prim_dist(const rmo_prim *pr, f3 p) {
    f3 pos = v3(pr->pos[0], pr->pos[1], pr->pos[2]);
    f4 q = {pr->quat[0], pr->quat[1], pr->quat[2], pr->quat[3]};
    f3 s = v3(pr->scale[0], pr->scale[1], pr->scale[2]);
    f3 pl = div3(rotate_q(sub3(p, pos), q), s);
    float d;
    switch (pr->type) {
    case RMO_SPHERE:   d = sd_sphere(pl, pr->params[0]); break;
    case RMO_BOX:      d = sd_box(pl, v3(pr->params[0], pr->params[1], pr->params[2])); break;
    case RMO_LINK:     d = sd_link(pl, pr->params[0], pr->params[1], pr->params[2]); break;
    case RMO_CYLINDER: d = sd_cylinder(pl, pr->params[0], pr->params[1]); break;
    default:           d = sd_torus(pl, pr->params[0], pr->params[1]); break;
    }
    d = d * min3(s);
    return (pr->op & RMO_OP_NEGATE) ? -d : d;
}

/* ---- cginc:51-60  rotate(float3 v, float3 a): euler angles, mul(mz, mul(my, mul(mx, v))) ----
 * cos/sin of the uniform angles are taken once per frame as the double-precision libm value
 * rounded to float.  The constant 0 and 1 entries of the three matrices are folded
 * (1*x + 0*y + 0*z -> x), which can only change the sign of a zero result; every caller feeds
 * the result to s*x - c, where that sign is lost. */
typedef struct { float cx, sx, cy, sy, cz, sz; } euler_t;

static euler_t euler_of(const float a[3]) {
    euler_t e;
    e.cx = (float)cos((double)a[0]); e.sx = (float)sin((double)a[0]);
    e.cy = (float)cos((double)a[1]); e.sy = (float)sin((double)a[1]);
    e.cz = (float)cos((double)a[2]); e.sz = (float)sin((double)a[2]);
    return e;
}

static inline f3 rotate_euler(f3 v, const euler_t *e) {
    f3 a = v3(v.x, e->cx * v.y + (-e->sx) * v.z, e->sx * v.y + e->cx * v.z);          /* mx */
    f3 b = v3(e->cy * a.x + e->sy * a.z, a.y, (-e->sy) * a.x + e->cy * a.z);          /* my */
    return v3(e->cz * b.x + (-e->sz) * b.y, e->sz * b.x + e->cz * b.y, b.z);          /* mz */
}

/* ---- canonical transcendental functions ----------------------------------------------
 * HLSL leaves sin/cos/acos/asin/atan2/pow/log precision to the driver, so the fractal bodies that
 * use them (cginc:194-262) have no bit-defined result in the reference.  They are fixed here as
 * float -> float functions evaluated in IEEE double with +,-,*,/ and sqrt only (no library call,
 * no fused multiply-add), in the operation order written below; the double result is rounded to
 * float once.  Any IEEE machine computes the same bits, and each result is within 1 float ulp of
 * the true function (tests/test_oracle_kat.py checks them against libm). */
static const double RM_PIO2_HI = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
static const double RM_PIO2_LO = 6.07710050650619224932e-11;   /* pi/2 - RM_PIO2_HI */
static const double RM_PI      = 3.14159265358979311600e+00;
static const double RM_PIO2    = 1.57079632679489655800e+00;
static const double RM_PIO4    = 7.85398163397448278999e-01;
static const double RM_LN2     = 6.93147180559945286227e-01;
static const double RM_LN2_HI  = 6.93147180369123816490e-01;   /* first 32 bits of ln 2 */
static const double RM_LN2_LO  = 1.90821492927058770002e-10;   /* ln 2 - RM_LN2_HI */

/* r in [-pi/4, pi/4]: Taylor series of sin and cos in Horner form on r*r */
static inline double c_sin_poly(double r) {
    double q = r * r;
    double p = 1.0 / 355687428096000.0;                  /*  1/17! */
    p = p * q - 1.0 / 1307674368000.0;                   /* -1/15! */
    p = p * q + 1.0 / 6227020800.0;                      /*  1/13! */
    p = p * q - 1.0 / 39916800.0;                        /* -1/11! */
    p = p * q + 1.0 / 362880.0;                          /*  1/9!  */
    p = p * q - 1.0 / 5040.0;                            /* -1/7!  */
    p = p * q + 1.0 / 120.0;                             /*  1/5!  */
    p = p * q - 1.0 / 6.0;                               /* -1/3!  */
    return r + r * (q * p);
}
static inline double c_cos_poly(double r) {
    double q = r * r;
    double p = 1.0 / 20922789888000.0;                   /*  1/16! */
    p = p * q - 1.0 / 87178291200.0;                     /* -1/14! */
    p = p * q + 1.0 / 479001600.0;                       /*  1/12! */
    p = p * q - 1.0 / 3628800.0;                         /* -1/10! */
    p = p * q + 1.0 / 40320.0;                           /*  1/8!  */
    p = p * q - 1.0 / 720.0;                             /* -1/6!  */
    p = p * q + 1.0 / 24.0;                              /*  1/4!  */
    p = p * q - 0.5;                                     /* -1/2!  */
    return 1.0 + q * p;
}
/* quadrant k and reduced argument: x = k*pi/2 + r */
static inline double c_reduce(double x, int *quadrant) {
    double k = floor(x * (1.0 / RM_PIO2) + 0.5);
    double r = (x - k * RM_PIO2_HI) - k * RM_PIO2_LO;
    double k4 = k - 4.0 * floor(k * 0.25);
    *quadrant = (int)k4;
    return r;
}
static inline float c_sin(float xf) {
    int n;
    double r = c_reduce((double)xf, &n);
    double v = (n & 1) ? c_cos_poly(r) : c_sin_poly(r);
    return (float)((n & 2) ? -v : v);
}
static inline float c_cos(float xf) {
    int n;
    double r = c_reduce((double)xf, &n);
    double v = (n & 1) ? c_sin_poly(r) : c_cos_poly(r);
    return (float)(((n + 1) & 2) ? -v : v);
}
/* atan of t in [0, 1] */
static inline double c_atan01(double t) {
    double base = 0.0, u = t;
    if (t > 0.41421356237309503) { u = (t - 1.0) / (t + 1.0); base = RM_PIO4; }      /* tan(pi/8) */
    double q = u * u;
    double p = 1.0 / 43.0;
    for (int n = 41; n >= 1; n -= 2) p = 1.0 / (double)n - q * p;                     /* 1 - q/3 + q^2/5 - ... */
    return base + u * p;
}
static inline double c_atan2d(double y, double x) {
    double ay = fabs(y), ax = fabs(x);
    double a;
    if (ay > ax) a = RM_PIO2 - c_atan01(ax / ay);
    else if (ax > 0.0) a = c_atan01(ay / ax);
    else a = ay / ax;                       /* 0/0 (and any NaN) stays NaN: atan2(0,0) is undefined in HLSL */
    if (x < 0.0) a = RM_PI - a;
    return (y < 0.0) ? -a : a;
}
static inline float c_atan2(float y, float x) { return (float)c_atan2d((double)y, (double)x); }
static inline float c_acos(float vf) {
    double v = (double)vf, t = 1.0 - v * v;
    if (!(t > 0.0)) t = 0.0;
    return (float)c_atan2d(sqrt(t), v);
}
static inline float c_asin(float vf) {
    double v = (double)vf, t = 1.0 - v * v;
    if (!(t > 0.0)) t = 0.0;
    return (float)c_atan2d(v, sqrt(t));
}
/* natural log of a finite x > 0 */
static inline double c_logd(double x) {
    int e;
    double m = frexp(x, &e) * 2.0;          /* m in [1, 2), x = m * 2^(e-1) */
    e -= 1;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0), q = s * s;
    double p = 1.0 / 25.0;
    for (int n = 23; n >= 1; n -= 2) p = 1.0 / (double)n + q * p;                     /* 1 + q/3 + q^2/5 + ... */
    return (double)e * RM_LN2 + 2.0 * s * p;
}
static inline double c_expd(double y) {
    if (!(y < 700.0)) return y > 0.0 ? INFINITY : y;        /* overflow; NaN stays NaN */
    if (y < -700.0) return 0.0;
    double k = floor(y * (1.0 / RM_LN2) + 0.5);
    double f = (y - k * RM_LN2_HI) - k * RM_LN2_LO;
    double p = 1.0 / 87178291200.0;                                                   /* 1/14! */
    static const double inv_fact[14] = {1.0, 1.0, 1.0 / 2.0, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0,
                                        1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0,
                                        1.0 / 479001600.0, 1.0 / 6227020800.0};
    for (int n = 13; n >= 0; n--) p = inv_fact[n] + f * p;
    return ldexp(p, (int)k);
}
static inline float c_log(float xf) {
    if (!(xf > 0.0f)) return xf == 0.0f ? -INFINITY : NAN;
    if (isinf(xf)) return xf;
    return (float)c_logd((double)xf);
}
static inline float c_pow(float xf, float ef) {             /* x >= 0 only (a length) */
    if (!(xf > 0.0f)) {
        if (xf == 0.0f) return ef > 0.0f ? 0.0f : (ef == 0.0f ? 1.0f : INFINITY);
        return NAN;
    }
    if (isinf(xf)) return ef > 0.0f ? xf : (ef == 0.0f ? 1.0f : 0.0f);
    return (float)c_expd((double)ef * c_logd((double)xf));
}

/* ---- cginc:194-227  mandelbulb(p, e, iters=12, bailout=10) ---------------------------------- */
static inline f4 mandelbulb(f3 p, float e) {
    f3 z = p;
    float dr = 1.0f, r = 0.0f;
    const float bailout = 10.0f;
    float o = bailout, o2 = bailout, o3 = bailout;
    for (int i = 0; i < 12; i++) {
        r = len3(z);
        o  = fminf(o,  len3(v3(z.x, z.y, z.z)));
        o2 = fminf(o2, len3(v3(z.x, z.y, z.z - 1.0f)));
        o3 = fminf(o3, len3(v3(z.x, z.y, z.z - 2.0f)));
        if (r > bailout) break;
        float theta = c_acos(z.z / r);
        float phi = c_atan2(z.y, z.x);
        dr = c_pow(r, e - 1.0f) * e * dr + 1.0f;
        float zr = c_pow(r, e);
        theta = theta * e;
        phi = phi * e;
        z = v3(zr * (c_sin(theta) * c_cos(phi)), zr * (c_sin(phi) * c_sin(theta)), zr * c_cos(theta));
        z = add3(z, p);
    }
    f4 out = {o, o2, o3, 0.5f * c_log(r) * r / dr};
    return out;
}

/* ---- cginc:229-262  mandelbrot(p, e, iters=12, bailout=10) ---------------------------------- */
static inline f4 mandelbrot3(f3 p, float e) {
    f3 z = p;
    float dr = 1.0f, r = 0.0f;
    const float bailout = 10.0f;
    float o = bailout, o2 = bailout, o3 = bailout;
    for (int i = 0; i < 12; i++) {
        o  = fminf(o,  len3(v3(z.x, z.y, z.z)));
        o2 = fminf(o2, len3(v3(z.x - 1.0f, z.y, z.z)));
        o3 = fminf(o3, len3(v3(z.x - 2.0f, z.y, z.z)));
        r = len3(z);
        if (r > bailout) break;
        float theta = c_atan2(z.y, z.x);
        float phi = c_asin(z.z / r);
        dr = c_pow(r, e - 1.0f) * e * dr + 1.0f;
        float zr = c_pow(r, e);
        theta = theta * e;
        phi = phi * e;
        z = v3(zr * (c_cos(theta) * c_cos(phi)), zr * (c_sin(theta) * c_cos(phi)), zr * (-c_sin(phi)));
        z = add3(z, p);
    }
    f4 out = {o, o2, o3, 0.5f * c_log(r) * r / dr};
    return out;
}

/* ---- cginc:295-313  menger(p): rgb = orbit traps, w = distance estimate -------------------- */
static inline f4 menger(f3 p, const euler_t *rot, float pow3) {
    float x = p.x, y = p.y, z = p.z, t;
    float o = 50.0f, o2 = 50.0f, o3 = 50.0f;
    for (int n = 0; n < 12; n++) {
        x = fabsf(x); y = fabsf(y); z = fabsf(z);
        if (x < y) { t = x; x = y; y = t; }
        if (y < z) { t = y; y = z; z = t; }
        if (x < y) { t = x; x = y; y = t; }
        f3 q = rotate_euler(v3(x, y, z), rot);
        x = q.x; y = q.y; z = q.z;
        x = x * 3.0f - 2.0f; y = y * 3.0f - 2.0f; z = z * 3.0f - 2.0f;
        if (z < -1.0f) z += 2.0f;
        o  = fminf(o,  len3(v3(x, y, z)));
        o2 = fminf(o2, len3(v3(x, y - 0.5f, z)));
        o3 = fminf(o3, len3(v3(x, y, z - 0.5f)));
    }
    f4 r = {o, o2, o3, (sqrtf(x * x + y * y + z * z) - 1.5f) * pow3};       /* pow3 = pow(3.0, -12) */
    return r;
}

/* ---- cginc:264-293  sierpinski(p) ----------------------------------------------------------
 * pows[i] = pow(scale, -i) for the 16 values the loop counter can end on */
static inline f4 sierpinski(f3 p, float scale, const euler_t *rot, const float pows[16]) {
    float x = p.x, y = p.y, z = p.z;
    float r = x * x + y * y + z * z;
    const float bailout = 20.0f;
    float o = bailout, o2 = bailout, o3 = bailout;
    int i;
    for (i = 0; i < 15 && r < bailout; i++) {
        if (x + y < 0.0f) { float x1 = -y; y = -x; x = x1; }
        if (x + z < 0.0f) { float x1 = -z; z = -x; x = x1; }
        if (y + z < 0.0f) { float y1 = -z; z = -y; y = y1; }
        f3 c = rotate_euler(v3(x, y, z), rot);
        x = c.x; y = c.y; z = c.z;
        x = scale * x - (scale - 1.0f);
        y = scale * y - (scale - 1.0f);
        z = scale * z - (scale - 1.0f);
        r = x * x + y * y + z * z;
        o  = fminf(o,  len3(v3(x - 1.0f, y, z)));
        o2 = fminf(o2, len3(v3(x, y - 1.0f, z)));
        o3 = fminf(o3, len3(v3(x, y, z - 1.0f)));
    }
    f4 out = {o, o2, o3, (sqrtf(r) - 2.0f) * pows[i]};
    return out;
}

/* ---- cginc:125-127  rand: frac(sin(dot(co, float2(12.9898,78.233))) * 43758.5453), with sin the canonical c_sin and
 * frac(x) = x - floor(x) ------------------------------------------------------------------------------------- */
static inline float frac1(float x) { return x - floorf(x); }
static inline float rand2(float x, float y) {
    return frac1(c_sin(x * 12.9898f + y * 78.233f) * 43758.5453f);
}
/* ---- cginc:129-149  noise: value noise, cubic Hermite interpolation of the four corner values ---------------- */
static inline float noise2(float x, float y) {
    float ix = floorf(x), iy = floorf(y);
    float fx = frac1(x), fy = frac1(y);
    float a = rand2(ix, iy), b = rand2(ix + 1.0f, iy + 0.0f), c = rand2(ix + 0.0f, iy + 1.0f), d = rand2(ix + 1.0f, iy + 1.0f);
    float ux = fx * fx * (3.0f - 2.0f * fx), uy = fy * fy * (3.0f - 2.0f * fy);
    return lerp1(a, b, ux) + (c - a) * uy * (1.0f - ux) + (d - b) * ux * uy;
}
/* ---- cginc:315-330  terrain(float2 x).w.  `float2 n = noise(p)` broadcasts the scalar; d accumulates in .x only ---- */
static inline float terrain_w(float x, float y) {
    float px = x * 0.003f / 250.0f, py = y * 0.003f / 250.0f;
    float a = 0.0f, b = 1.0f, dx = 0.0f, dy = 0.0f;
    for (int i = 0; i < 3; i++) {
        float n = noise2(px, py);
        dx += n; dy += 0.0f;
        a += b * n / (1.0f + (dx * dx + dy * dy));
        b *= 0.5f;
        px = .8f * px * 2.0f; py = -.6f * py * 2.0f;
    }
    return 250.0f * 120.0f * a;
}

/* atan(x) of getscatter (shader:139) as the canonical atan2(x, 1) */
static inline float c_atan(float x) { return c_atan2(x, 1.0f); }

/* ---- shader:84-100  getbrightnesshard / getbrightness with the light of shader:157-162 (range 1000, intensity 1) ---- */
static inline float getbrightness(f3 p, f3 n, f3 lpos, int hard) {
    const float range = 1000.0f, intensity = 1.0f;
    float dist = len3(sub3(p, lpos));
    if (dist >= range) return 0.0f;
    float dn = dot3(n, sub3(lpos, p));
    if (hard) return lerp1(clamp1(dn, 0.0f, 1.0f) * intensity, 0.0f, dist / range);
    float d = clamp1(dn, -1.0f, 1.0f) * .5f + .5f;
    d *= d;
    return lerp1(d * intensity, 0.0f, dist / range);
}
/* ---- shader:129-140  getscatter(ray r, light l) --------------------------------------------------------------- */
static inline float getscatter(f3 ro, f3 rd, float length, f3 lpos) {
    const float intensity = 1.0f;
    f3 q = sub3(ro, lpos);
    float b = dot3(rd, q), c = dot3(q, q);
    float s = intensity / sqrtf(c - b * b);
    return s * (c_atan((length + b) * s) - c_atan(b * s));
}
/* ---- shader:142-147  refract2 ---------------------------------------------------------------------------------- */
static inline f3 refract2(f3 i, f3 n, float eta) {
    eta = 2.0f - eta;
    float cosi = dot3(n, i);
    return sub3(scl3(i, eta), scl3(n, -cosi + eta * cosi));
}

/* the reference's fixed scene as a primitive table: sphere, box, link(5,5,.1),
 * cylinder with colours (1,.25,.25) (.25,1,.25) (.25,.25,1) (1,1,.25)
 * (cginc:351-359); positions/rotations/scales from the three matrices */
static void legacy_prims(const rmo_uniforms *u, rmo_prim out[4]) {
    static const float col[4][3] = {{1, .25f, .25f}, {.25f, 1, .25f}, {.25f, .25f, 1}, {1, 1, .25f}};
    static const float par[4][4] = {{1, 0, 0, 0}, {1, 1, 1, 0}, {5, 5, .1f, 0}, {.5f, .5f, 0, 0}};
    for (int i = 0; i < 4; i++) {
        f4 p = mrow(u->positions, i), q = mrow(u->rotations, i), s = mrow(u->scales, i);
        out[i].type = i;
        out[i].op = RMO_OP_SMIN;
        out[i].pos[0] = p.x; out[i].pos[1] = p.y; out[i].pos[2] = p.z;
        out[i].quat[0] = q.x; out[i].quat[1] = q.y; out[i].quat[2] = q.z; out[i].quat[3] = q.w;
        out[i].scale[0] = s.x; out[i].scale[1] = s.y; out[i].scale[2] = s.z;
        memcpy(out[i].color, col[i], sizeof col[i]);
        memcpy(out[i].params, par[i], sizeof par[i]);
    }
}

typedef struct {
    const rmo_uniforms *u;
    const rmo_prim *prims;
    int n;
    rmo_prim legacy[4];
    f3 ro, light;
    f3 tl, tr, br, bl;     /* world-space corner rays */
    int body;              /* RMO_BODY_* */
    euler_t frot;          /* cos/sin of _FractalRotationX/Y/Z */
    float pow3;            /* pow(3.0, -12)            (cginc:312) */
    float pows[16];        /* pow(_FractalScale, -i)   (cginc:292) */
    float fpower;          /* e of mandelbulb (default 7, cginc:194) / mandelbrot (default 2, cginc:229) */
    euler_t rot_time;      /* cos/sin of float3(0, 0, _Time.y)                                  (cginc:349) */
    euler_t rot_menger;    /* cos/sin of float3(_Time.y*.25, _Time.y*1.4*.25, _Time.y*1.6*.25)  (cginc:349) */
} scene_ctx;

/* cginc:349, the body scene() returns when the commented line is live: the world-space p, no wrapper */
static inline f4 scene_composite(const scene_ctx *c, f3 p) {
    float e = menger(rotate_euler(p, &c->rot_menger), &c->frot, c->pow3).w + 1.1f;
    return mandelbulb(rotate_euler(p, &c->rot_time), e);
}

/* fractal scene(): the per-primitive wrapper of cginc:336-354 with slot 0, around a fractal body */
static inline f4 scene_fractal(const scene_ctx *c, f3 p) {
    if (c->body == RMO_BODY_COMPOSITE) return scene_composite(c, p);
    const rmo_prim *pr = &c->legacy[0];
    f3 pos = v3(pr->pos[0], pr->pos[1], pr->pos[2]);
    f4 q = {pr->quat[0], pr->quat[1], pr->quat[2], pr->quat[3]};
    f3 s = v3(pr->scale[0], pr->scale[1], pr->scale[2]);
    f3 pl = div3(rotate_q(sub3(p, pos), q), s);
    f4 o;
    switch (c->body) {
    case RMO_BODY_MENGER:     o = menger(pl, &c->frot, c->pow3); break;
    case RMO_BODY_SIERPINSKI: o = sierpinski(pl, c->u->fractal_scale, &c->frot, c->pows); break;
    case RMO_BODY_MANDELBULB: o = mandelbulb(pl, c->fpower); break;
    case RMO_BODY_TERRAIN:    { f4 t = {1.0f, 1.0f, 1.0f, pl.y - terrain_w(pl.x, pl.z)}; o = t; } break;
    default:                  o = mandelbrot3(pl, c->fpower); break;
    }
    o.w = o.w * min3(s);
    return o;
}

/* ---- cginc:84-99  mod / shmod.  HLSL fmod is the remainder with the sign of x and |f| < |m|; it is
 * taken here as C's fmodf, which is that remainder exactly (the shader compiler's x - m*trunc(x/m)
 * expansion is not bit-defined). ------------------------------------------------------------ */
static inline float mod1(float x, float m) {
    float r = fmodf(x, m);
    return r < 0.0f ? r + m : r;
}
static inline float shmod1(float x, float m) { return mod1(x + m * .5f, m) - m * .5f; }
/* ---- cginc:156-159  boxfold ------------------------------------------------------------------ */
static inline f3 boxfold(f3 p, float b) {
    b = b * .5f;
    return v3(clamp1(p.x, -b, b) * 2.0f - p.x, clamp1(p.y, -b, b) * 2.0f - p.y, clamp1(p.z, -b, b) * 2.0f - p.z);
}
/* ---- cginc:161-165  spherefold --------------------------------------------------------------- */
static inline f3 spherefold(f3 p, float R) {
    float r = len3(p);
    if (r < R) {
        float rr = r * r;
        p = v3(p.x * R * R / rr, p.y * R * R / rr, p.z * R * R / rr);
    }
    return p;
}
static inline f3 domain(const rmo_uniforms *u, f3 p) {
    switch (u->domain_op) {
    case 1:
        if (u->domain_period[0] > 0.0f) p.x = shmod1(p.x, u->domain_period[0]);
        if (u->domain_period[1] > 0.0f) p.y = shmod1(p.y, u->domain_period[1]);
        if (u->domain_period[2] > 0.0f) p.z = shmod1(p.z, u->domain_period[2]);
        break;
    case 2: p = boxfold(p, u->domain_size); break;
    case 3: p = spherefold(p, u->domain_size); break;
    default: break;
    }
    return p;
}

#define RMO_MAX_PRIMS 1024

/* ---- cginc:334-364  scene(p): float4(rgb, distance) ----------------------
 * For n == 4 this is literally smin(smin(o0,o1), smin(o2,o3)) (cginc:363).  For
 * other n the same balanced pairwise tree is built level by level in index
 * order; an odd element at the end of a level is carried up unchanged. */
static inline f4 scene(const scene_ctx *c, f3 p) {
    p = domain(c->u, p);
    if (c->body != RMO_BODY_PRIMS) return scene_fractal(c, p);
    f4 o[RMO_MAX_PRIMS];
    int first[RMO_MAX_PRIMS];          /* leftmost leaf under each node: its op bits choose the merge */
    int n = c->n;
    for (int i = 0; i < n; i++) {
        const rmo_prim *pr = &c->prims[i];
        o[i].w = prim_dist(pr, p);
        o[i].x = pr->color[0]; o[i].y = pr->color[1]; o[i].z = pr->color[2];
        first[i] = i;
    }
    float k = c->u->smooth_k;
    while (n > 1) {
        int m = 0;
        for (int i = 0; i + 1 < n; i += 2) {
            int smax = c->prims[first[i + 1]].op & RMO_OP_SMAX;
            first[m] = first[i];
            o[m++] = smax ? smax4(o[i], o[i + 1], k) : smin4(o[i], o[i + 1], k);
        }
        if (n & 1) { first[m] = first[n - 1]; o[m++] = o[n - 1]; }
        n = m;
    }
    return o[0];
}

/* distance-only evaluation: exactly scene(p).w -- the .w lane of smin4 never
 * reads rgb, so dropping the colour blend changes no bit of the result */
static inline float scene_w(const scene_ctx *c, f3 p) {
    p = domain(c->u, p);
    if (c->body != RMO_BODY_PRIMS) return scene_fractal(c, p).w;
    float o[RMO_MAX_PRIMS];
    int first[RMO_MAX_PRIMS];
    int n = c->n;
    for (int i = 0; i < n; i++) { o[i] = prim_dist(&c->prims[i], p); first[i] = i; }
    float k = c->u->smooth_k;
    while (n > 1) {
        int m = 0;
        for (int i = 0; i + 1 < n; i += 2) {
            float a = o[i], b = o[i + 1];
            first[m] = first[i];
            if (c->prims[first[i + 1]].op & RMO_OP_SMAX) {
                float h = clamp1(0.5f + 0.5f * (a - b) / k, 0.0f, 1.0f);
                o[m++] = -lerp1(-b, -a, h) - k * h * (1.0f - h);
            } else {
                float h = clamp1(0.5f + 0.5f * (b - a) / k, 0.0f, 1.0f);
                o[m++] = lerp1(b, a, h) - k * h * (1.0f - h);
            }
        }
        if (n & 1) { first[m] = first[n - 1]; o[m++] = o[n - 1]; }
        n = m;
    }
    return o[0];
}

typedef struct { int hit; int evals; float length; float contact; int steps; /* r.steps: the loop counter after the loop (cginc:387) */ } march_t;

/* ---- cginc:367-392  raymarch(ro, rd, depth, steps=2000) -------------------
 * `hit` is read uninitialised in the reference when the loop never sets it;
 * it is false here.  `contact` models the shader-global _ContactThreshold that
 * the loop overwrites every iteration (cginc:376). */
static inline march_t raymarch(const scene_ctx *c, f3 ro, f3 rd, float depth, float contact0) {
    const rmo_uniforms *u = c->u;
    depth = fminf(depth, u->max_dist);
    float steps = fminf(2000.0f, u->max_steps);
    float dm = 0.0f;
    march_t r = {0, 0, 0.0f, contact0, 0};
    int i;
    for (i = 0; (float)i < steps; i++) {
        f3 cp = add3(ro, scl3(rd, dm));
        float dts = scene_w(c, cp);
        r.evals++;
        dm += fabsf(dts) * u->step_factor;
        r.contact = dm * .0025f;
        if (fabsf(dts) < r.contact) { r.hit = 1; break; }
        if (dm > depth) break;
    }
    r.length = dm;
    r.steps = i;
    return r;
}

/* ---- shader:50-68  getnormalraw(p, s=0): central differences, NOT divided -- */
static inline f3 getnormalraw(const scene_ctx *c, f3 p) {
    float e = fmaxf(0.0f, c->u->normal_sample_scale);
    f3 ex = v3(e, 0, 0), ey = v3(0, e, 0), ez = v3(0, 0, e);
    return v3(scene_w(c, add3(p, ex)) - scene_w(c, sub3(p, ex)),
              scene_w(c, add3(p, ey)) - scene_w(c, sub3(p, ey)),
              scene_w(c, add3(p, ez)) - scene_w(c, sub3(p, ez)));
}

/* ---- shader:102-123  getshadow(p, n, k=16, depth=2000, steps=2000) --------
 * reads the contact threshold left behind by raymarch() (shader:107); the first
 * iteration divides by dm == 0 (shader:111) */
static inline float getshadow(const scene_ctx *c, f3 p, f3 n, float contact, int *evals) {
    const rmo_uniforms *u = c->u;
    const float k = 16.0f;
    float depth = fminf(2000.0f, u->max_dist);
    float steps = fminf(2000.0f, u->max_steps);
    float dm = 0.0f, res = 1.0f;
    p = add3(p, scl3(scl3(n, contact), 2.0f));
    int i, ev = 0;
    for (i = 0; (float)i < steps; i++) {
        f3 cp = add3(p, scl3(c->light, dm));
        float dts = scene_w(c, cp);
        ev++;
        res = fminf(res, k * dts / dm);
        dm += fabsf(dts) * u->step_factor;
        contact = dm * .0025f;
        if (fabsf(dts) < contact) break;
        if (dm > depth) break;
    }
    *evals = ev;
    return res;
}

/* Shadow-map lookup in place of the shadow march (definition in rm_oracle.h; the reference declares _ShadowMap and
 * _ShadowMapVPMatrix at shader:45-46 and never reads them).  Returns 1 (lit) or 0 (in shadow). */
static inline float shadow_lookup(const scene_ctx *c, f3 p, f3 n, float contact) {
    const rmo_uniforms *u = c->u;
    f3 q = add3(p, scl3(scl3(n, contact), 2.0f));                    /* shader:107 */
    float uu, vv, d;
    if (u->shadow_mode == 1) {
        f4 r0 = mrow(u->shadow_vp, 0), r1 = mrow(u->shadow_vp, 1), r3 = mrow(u->shadow_vp, 3);
        float cx = ((r0.x * q.x + r0.y * q.y) + r0.z * q.z) + r0.w;
        float cy = ((r1.x * q.x + r1.y * q.y) + r1.z * q.z) + r1.w;
        float cw = ((r3.x * q.x + r3.y * q.y) + r3.z * q.z) + r3.w;
        if (!(cw > 0.0f)) return 1.0f;
        uu = (cx / cw) * .5f + .5f;
        vv = (cy / cw) * .5f + .5f;
        d = cw;
    } else {
        const float *k = u->shadow_corners;
        f3 bl = v3(k[9], k[10], k[11]);
        f3 e1 = sub3(v3(k[6], k[7], k[8]), bl), e2 = sub3(v3(k[0], k[1], k[2]), bl);
        f3 r = sub3(q, bl);
        uu = dot3(r, e1) / dot3(e1, e1);
        vv = dot3(r, e2) / dot3(e2, e2);
        d = dot3(r, v3(u->shadow_forward[0], u->shadow_forward[1], u->shadow_forward[2]));
    }
    if (!(uu >= 0.0f && uu < 1.0f && vv >= 0.0f && vv < 1.0f)) return 1.0f;
    int ix = (int)floorf(uu * (float)u->shadow_w), iy = (int)floorf(vv * (float)u->shadow_h);
    if (ix > u->shadow_w - 1) ix = u->shadow_w - 1;
    if (iy > u->shadow_h - 1) iy = u->shadow_h - 1;
    float stored = u->shadow_map[(size_t)iy * (size_t)u->shadow_w + (size_t)ix];
    return d > stored * (1.0f + u->shadow_bias) ? 0.0f : 1.0f;
}

/* ---- shader:125-127  getAO ------------------------------------------------ */
static inline float getAO(const scene_ctx *c, f3 n) {
    float s = c->u->normal_sample_scale;
    return clamp1(len3(v3(n.x / s, n.y / s, n.z / s)), 0.0f, 1.0f);
}

/* ---- cginc:26-33  vert: corner select ------------------------------------- */
int rmo_vert_corner_index(float uvx, float uvy) {
    return (int)(uvx + (uvx != 0.0f ? 1.0f - uvy : (1.0f - uvy) * 3.0f));
}

/* mul(_CameraInvViewMatrix, corner.xyz): float4x4 * float3 uses the upper-left
 * 3x3 (cginc:31) */
static inline f3 mul33(const float *m, f3 v) {
    f4 r0 = mrow(m, 0), r1 = mrow(m, 1), r2 = mrow(m, 2);
    return v3(dot3(xyz(r0), v), dot3(xyz(r1), v), dot3(xyz(r2), v));
}

static void scene_ctx_init(scene_ctx *c, const rmo_uniforms *u) {
    c->u = u;
    legacy_prims(u, c->legacy);
    if (u->n_prims > 0) { c->prims = u->prims; c->n = u->n_prims; }
    else { c->prims = c->legacy; c->n = 4; }
    c->body = u->scene_body;
    c->frot = euler_of(u->fractal_rotation);
    {
        const float t = u->time[1];
        if (u->animate_fractal) {                                       /* cginc:346-348 */
            float a[3] = {(float)sin((double)t) * .5f, (float)sin((double)(t * 1.414f)) * .5f, (float)sin((double)(t * 1.618f)) * .5f};
            c->frot = euler_of(a);
        }
        float at[3] = {0.0f, 0.0f, t};
        float am[3] = {t * .25f, t * 1.4f * .25f, t * 1.6f * .25f};
        c->rot_time = euler_of(at);
        c->rot_menger = euler_of(am);
    }
    c->pow3 = (float)pow(3.0, -12.0);
    c->fpower = u->fractal_power > 0.0f ? u->fractal_power : (u->scene_body == RMO_BODY_MANDELBULB ? 7.0f : 2.0f);
    for (int i = 0; i < 16; i++) c->pows[i] = (float)pow((double)u->fractal_scale, -(double)i);
    c->ro = v3(u->cam_pos[0], u->cam_pos[1], u->cam_pos[2]);
    c->light = v3(u->light[0], u->light[1], u->light[2]);   /* float4 -> float3 truncation */
    c->tl = mul33(u->cam_inv_view, xyz(mrow(u->frustum_corners, 0)));
    c->tr = mul33(u->cam_inv_view, xyz(mrow(u->frustum_corners, 1)));
    c->br = mul33(u->cam_inv_view, xyz(mrow(u->frustum_corners, 2)));
    c->bl = mul33(u->cam_inv_view, xyz(mrow(u->frustum_corners, 3)));
}

/* rasteriser interpolation of viewDir to the pixel centre, uv origin bottom-left:
 * rd = lerp(lerp(BL,BR,u), lerp(TL,TR,u), v) */
static inline f3 pixel_ray(const scene_ctx *c, int x, int y, int w, int h) {
    float uu = ((float)x + 0.5f) / (float)w;
    float vv = ((float)y + 0.5f) / (float)h;
    return lerp3(lerp3(c->bl, c->br, uu), lerp3(c->tl, c->tr, uu), vv);
}

static inline uint8_t unorm8(float x) {
    x = fminf(fmaxf(x, 0.0f), 1.0f);      /* saturate; NaN -> 0 */
    return (uint8_t)floorf(x * 255.0f + 0.5f);
}

typedef struct {
    scene_ctx sc;
    const uint8_t *src; const float *sdepth;
    uint8_t *rgba; float *depth; uint8_t *hit; float *rgbf; uint16_t *sp, *ss;
    int w, h, y0, y1;
    int mode;                 /* 0 frag, 1 perspective depth, 2 directional depth */
    f3 dc[4]; f3 dfwd;        /* directional pass: corners TL,TR,BR,BL and forward */
    atomic_int next_row;
    pthread_mutex_t mu;
    rmo_stats st;
} job_t;

/* HLSL reflect / refract, expanded once (not bit-defined by D3D):
 *   reflect(i, n)      = i - (2*dot(i,n))*n
 *   refract(i, n, eta) = k < 0 ? 0 : eta*i - (eta*dot(n,i) + sqrt(k))*n,  k = 1 - eta*eta*(1 - dot(n,i)^2) */
static inline f3 hlsl_reflect(f3 i, f3 n) { return sub3(i, scl3(n, 2.0f * dot3(i, n))); }
static inline f3 hlsl_refract(f3 i, f3 n, float eta) {
    float d = dot3(n, i);
    float k = 1.0f - (eta * eta) * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f, 0.0f, 0.0f);
    return sub3(scl3(i, eta), scl3(n, eta * d + sqrtf(k)));
}

/* ---- shader:149-216  frag --------------------------------------------------
 * The fx_* uniforms switch on lines the reference keeps commented out: the one-bounce loop
 * (shader:178-187), the alternative outputs (:198-201), the step-count term (:203) and the light
 * integration along the ray (:206-212).  All zero = the shader as it stands. */
static void frag_pixel(job_t *j, int x, int y, rmo_stats *st) {
    const scene_ctx *c = &j->sc;
    const rmo_uniforms *u = c->u;
    size_t idx = (size_t)y * (size_t)j->w + (size_t)x;

    f3 tex;
    if (j->src) tex = v3(j->src[idx * 4] / 255.0f, j->src[idx * 4 + 1] / 255.0f, j->src[idx * 4 + 2] / 255.0f);
    else tex = v3(u->background[0], u->background[1], u->background[2]);
    float depth = j->sdepth ? j->sdepth[idx] : u->far_plane;          /* shader:154 */

    f3 ro = c->ro;
    f3 rd = pixel_ray(c, x, y, j->w, j->h);
    march_t r = raymarch(c, ro, rd, depth, u->contact_threshold);     /* shader:171 */
    st->sum_steps_p += (uint64_t)r.evals;

    f3 col = tex;
    int sev = 0;
    const march_t primary = r;                                        /* depth / mask outputs describe the camera ray */
    const int fx = u->fx_step_heat || u->fx_debug_view || u->fx_volumetric || u->fx_bounce || u->fx_scatter;
    if (r.hit || u->shade_miss_pixels || fx) {
        f3 hitpoint = add3(ro, scl3(rd, r.length));                   /* shader:174 */
        f3 rawnormal = getnormalraw(c, hitpoint);
        f3 normal = normalize3(rawnormal);
        if (u->fx_bounce && r.hit) {                                  /* shader:178-187, one round */
            ro = sub3(hitpoint, scl3(scl3(normal, r.contact), 2.0f));
            rd = scene_w(c, hitpoint) >= 0.0f ? (u->fx_bounce == 2 ? refract2(rd, normal, u->refractive_index) : hlsl_refract(rd, normal, 1.1f))
                                              : hlsl_reflect(rd, normal);
            r = raymarch(c, ro, rd, 2000.0f, r.contact);
            hitpoint = add3(ro, scl3(rd, r.length));
            rawnormal = getnormalraw(c, hitpoint);
            normal = normalize3(rawnormal);
        }
        f4 s = scene(c, hitpoint);                                    /* shader:193 */
        f3 srgb = xyz(s);
        float sh = u->shadow_mode != 0 ? shadow_lookup(c, hitpoint, normal, r.contact) : getshadow(c, hitpoint, normal, r.contact, &sev);
        float shadow = remap(sh, 0.0f, 1.0f, .25f, 1.0f);
        float ndl = dot3(normal, c->light) * .5f + .5f;
        if (u->fx_light_model)
            ndl = getbrightness(hitpoint, normal, v3(u->light_position[0], u->light_position[1], u->light_position[2]), u->fx_light_model == 2);
        f3 shaded = scl3(scl3(scl3(srgb, ndl), shadow), getAO(c, rawnormal));
        if (u->fx_debug_view == 1) shaded = v3(shadow, shadow, shadow);                                    /* shader:200 */
        if (u->fx_debug_view == 2) shaded = v3(normal.x * .5f + .5f, normal.y * .5f + .5f, normal.z * .5f + .5f);   /* :201 */
        if (u->fx_debug_view == 3) { float v = clamp1(s.x * 2.5f - 1.75f, 0.0f, 1.0f); shaded = v3(v, v, v); }      /* :198 */
        st->sum_steps_s_all += (uint64_t)sev;
        if (r.hit) col = shaded;
        if (primary.hit) { st->sum_steps_s += (uint64_t)sev; st->n_hit++; }
    }
    if (u->fx_step_heat) {                                            /* shader:203 */
        float h = (float)r.steps / 100.0f * u->step_factor;
        col = v3(col.x + h, col.y + h, col.z + h);
    }
    if (u->fx_scatter) {                                              /* shader:129-140, added like the light integration below */
        float sc = getscatter(ro, rd, r.length, v3(u->light_position[0], u->light_position[1], u->light_position[2]));
        col = v3(col.x + sc, col.y + sc, col.z + sc);
    }
    if (u->fx_volumetric) {                                           /* shader:206-212 */
        float la = 0.0f;
        const f3 zero = v3(0.0f, 0.0f, 0.0f);
        for (float i = 0.0f; i * .1f < fminf(5.0f, r.length); i += 1.0f) {
            f3 rp = add3(ro, scl3(scl3(rd, i), .1f));
            int ev;
            la += getshadow(c, rp, zero, r.contact, &ev) * .05f;
        }
        col = v3(col.x + la, col.y + la, col.z + la);
    }
    r.length = primary.length; r.hit = primary.hit; r.evals = primary.evals;
    if (j->rgba) {
        j->rgba[idx * 4 + 0] = unorm8(col.x);
        j->rgba[idx * 4 + 1] = unorm8(col.y);
        j->rgba[idx * 4 + 2] = unorm8(col.z);
        j->rgba[idx * 4 + 3] = 255;                                   /* fixed4(col, 1) */
    }
    if (j->rgbf) { j->rgbf[idx * 3] = col.x; j->rgbf[idx * 3 + 1] = col.y; j->rgbf[idx * 3 + 2] = col.z; }
    if (j->depth) j->depth[idx] = r.length;
    if (j->hit) j->hit[idx] = (uint8_t)r.hit;
    if (j->sp) j->sp[idx] = (uint16_t)r.evals;
    if (j->ss) j->ss[idx] = (uint16_t)(r.hit ? sev : 0);
}

static void depth_pixel(job_t *j, int x, int y, rmo_stats *st) {
    const scene_ctx *c = &j->sc;
    size_t idx = (size_t)y * (size_t)j->w + (size_t)x;
    f3 ro, rd;
    if (j->mode == 1) {
        ro = c->ro;
        rd = pixel_ray(c, x, y, j->w, j->h);
    } else {
        float uu = ((float)x + 0.5f) / (float)j->w;
        float vv = ((float)y + 0.5f) / (float)j->h;
        ro = lerp3(lerp3(j->dc[3], j->dc[2], uu), lerp3(j->dc[0], j->dc[1], uu), vv);
        rd = j->dfwd;
    }
    march_t r = raymarch(c, ro, rd, c->u->far_plane, c->u->contact_threshold);
    st->sum_steps_p += (uint64_t)r.evals;
    st->n_hit += (uint64_t)r.hit;
    if (j->depth) j->depth[idx] = r.length;
    if (j->hit) j->hit[idx] = (uint8_t)r.hit;
}

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    rmo_stats st;
    memset(&st, 0, sizeof st);
    for (;;) {
        int y = atomic_fetch_add(&j->next_row, 1);
        if (y >= j->y1) break;
        for (int x = 0; x < j->w; x++) {
            if (j->mode == 0) frag_pixel(j, x, y, &st); else depth_pixel(j, x, y, &st);
            st.n_pixels++;
        }
    }
    pthread_mutex_lock(&j->mu);
    j->st.n_pixels += st.n_pixels; j->st.n_hit += st.n_hit;
    j->st.sum_steps_p += st.sum_steps_p; j->st.sum_steps_s += st.sum_steps_s;
    j->st.sum_steps_s_all += st.sum_steps_s_all;
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

int rmo_online_cpus(void) {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { int n = CPU_COUNT(&set); if (n > 0) return n; }
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

static int run_job(job_t *j, int threads, rmo_stats *stats) {
    if (j->sc.n > RMO_MAX_PRIMS || j->w <= 0 || j->h <= 0 || j->y0 < 0 || j->y1 > j->h) return -1;
    if (threads <= 0) threads = rmo_online_cpus();
    if (threads > 256) threads = 256;
    int rows = j->y1 - j->y0;
    if (threads > rows) threads = rows > 0 ? rows : 1;
    atomic_init(&j->next_row, j->y0);
    pthread_mutex_init(&j->mu, NULL);
    memset(&j->st, 0, sizeof j->st);
    pthread_t th[256];
    int started = 0;
    for (int t = 1; t < threads; t++)
        if (pthread_create(&th[started], NULL, worker, j) == 0) started++;
    worker(j);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&j->mu);
    if (stats) *stats = j->st;
    return 0;
}

int rmo_render(const rmo_uniforms *u, const uint8_t *src_rgba8, const float *scene_depth,
               uint8_t *dst_rgba8, float *dst_depth, uint8_t *dst_hit, float *dst_rgb_f32,
               uint16_t *dst_steps_p, uint16_t *dst_steps_s,
               int w, int h, int y0, int y1, int threads, rmo_stats *stats) {
    job_t *j = (job_t *)calloc(1, sizeof *j);
    if (!j) return -2;
    scene_ctx_init(&j->sc, u);
    j->src = src_rgba8; j->sdepth = scene_depth;
    j->rgba = dst_rgba8; j->depth = dst_depth; j->hit = dst_hit; j->rgbf = dst_rgb_f32;
    j->sp = dst_steps_p; j->ss = dst_steps_s;
    j->w = w; j->h = h; j->y0 = y0; j->y1 = y1; j->mode = 0;
    int rc = run_job(j, threads, stats);
    free(j);
    return rc;
}

/* Depth pass from a perspective shadow camera.  The reference's ShadowMap
 * fragment is a colour-inverting stub (Assets/ShadowMap.shader:20-25), so no
 * reference arithmetic exists beyond the bindings ShadowMap.Render() makes
 * (Assets/ShadowMap.cs:50-56): this pass is raymarch() along the rays those two
 * matrices define, writing r.length. */
int rmo_shadowmap_perspective(const rmo_uniforms *u, float *dst_depth, uint8_t *dst_hit,
                              int w, int h, int threads, rmo_stats *stats) {
    job_t *j = (job_t *)calloc(1, sizeof *j);
    if (!j) return -2;
    scene_ctx_init(&j->sc, u);
    j->depth = dst_depth; j->hit = dst_hit;
    j->w = w; j->h = h; j->y0 = 0; j->y1 = h; j->mode = 1;
    int rc = run_job(j, threads, stats);
    free(j);
    return rc;
}

/* Orthographic depth pass over the rectangle ShadowMapDirectional.Update()
 * computes (Assets/ShadowMapDirectional.cs:25-34, corner order TL,TR,BR,BL as
 * its initialCorners :11-16): origin bilinear over the corners, direction =
 * the transform's forward, marched with raymarch(). */
int rmo_shadowmap_directional(const rmo_uniforms *u, const float corners[12], const float forward[3],
                              float *dst_depth, uint8_t *dst_hit,
                              int w, int h, int threads, rmo_stats *stats) {
    job_t *j = (job_t *)calloc(1, sizeof *j);
    if (!j) return -2;
    scene_ctx_init(&j->sc, u);
    for (int i = 0; i < 4; i++) j->dc[i] = v3(corners[i * 3], corners[i * 3 + 1], corners[i * 3 + 2]);
    j->dfwd = v3(forward[0], forward[1], forward[2]);
    j->depth = dst_depth; j->hit = dst_hit;
    j->w = w; j->h = h; j->y0 = 0; j->y1 = h; j->mode = 2;
    int rc = run_job(j, threads, stats);
    free(j);
    return rc;
}

/* ---- probes for the known-answer tests ------------------------------------- */
float rmo_sphere(const float p[3], float r) { return sd_sphere(v3(p[0], p[1], p[2]), r); }
float rmo_box(const float p[3], const float b[3]) { return sd_box(v3(p[0], p[1], p[2]), v3(b[0], b[1], b[2])); }
float rmo_link(const float p[3], float le, float r1, float r2) { return sd_link(v3(p[0], p[1], p[2]), le, r1, r2); }
float rmo_cylinder(const float p[3], float h, float r) { return sd_cylinder(v3(p[0], p[1], p[2]), h, r); }
float rmo_torus(const float p[3], float tx, float ty) { return sd_torus(v3(p[0], p[1], p[2]), tx, ty); }
float rmo_mod(float x, float m) { return mod1(x, m); }
float rmo_shmod(float x, float m) { return shmod1(x, m); }
void rmo_boxfold(const float p[3], float b, float out[3]) { f3 r = boxfold(v3(p[0], p[1], p[2]), b); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void rmo_spherefold(const float p[3], float R, float out[3]) { f3 r = spherefold(v3(p[0], p[1], p[2]), R); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void rmo_rotate_q(const float v[3], const float q[4], float out[3]) {
    f4 qq = {q[0], q[1], q[2], q[3]};
    f3 r = rotate_q(v3(v[0], v[1], v[2]), qq);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float rmo_remap(float x, float o1, float o2, float n1, float n2) { return remap(x, o1, o2, n1, n2); }
float rmo_c_sin(float x) { return c_sin(x); }
float rmo_c_cos(float x) { return c_cos(x); }
float rmo_c_acos(float x) { return c_acos(x); }
float rmo_c_asin(float x) { return c_asin(x); }
float rmo_c_atan2(float y, float x) { return c_atan2(y, x); }
float rmo_c_log(float x) { return c_log(x); }
float rmo_c_pow(float x, float e) { return c_pow(x, e); }
float rmo_c_atan(float x) { return c_atan(x); }
float rmo_rand(const float co[2]) { return rand2(co[0], co[1]); }
float rmo_noise(const float st[2]) { return noise2(st[0], st[1]); }
float rmo_terrain(const float x[2]) { return terrain_w(x[0], x[1]); }
float rmo_getbrightness(const float p[3], const float n[3], const float lp[3], int hard) {
    return getbrightness(v3(p[0], p[1], p[2]), v3(n[0], n[1], n[2]), v3(lp[0], lp[1], lp[2]), hard);
}
float rmo_getscatter(const float ro[3], const float rd[3], float length, const float lp[3]) {
    return getscatter(v3(ro[0], ro[1], ro[2]), v3(rd[0], rd[1], rd[2]), length, v3(lp[0], lp[1], lp[2]));
}
void rmo_refract2(const float i[3], const float n[3], float eta, float out[3]) {
    f3 r = refract2(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]), eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void rmo_mandelbulb(const float p[3], float e, float out[4]) {
    f4 r = mandelbulb(v3(p[0], p[1], p[2]), e);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void rmo_mandelbrot(const float p[3], float e, float out[4]) {
    f4 r = mandelbrot3(v3(p[0], p[1], p[2]), e);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void rmo_menger(const rmo_uniforms *u, const float p[3], float out[4]) {
    euler_t e = euler_of(u->fractal_rotation);
    f4 r = menger(v3(p[0], p[1], p[2]), &e, (float)pow(3.0, -12.0));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void rmo_sierpinski(const rmo_uniforms *u, const float p[3], float out[4]) {
    euler_t e = euler_of(u->fractal_rotation);
    float pows[16];
    for (int i = 0; i < 16; i++) pows[i] = (float)pow((double)u->fractal_scale, -(double)i);
    f4 r = sierpinski(v3(p[0], p[1], p[2]), u->fractal_scale, &e, pows);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void rmo_smax4(const float a[4], const float b[4], float k, float out[4]) {
    f4 aa = {a[0], a[1], a[2], a[3]}, bb = {b[0], b[1], b[2], b[3]};
    f4 r = smax4(aa, bb, k);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void rmo_smin4(const float a[4], const float b[4], float k, float out[4]) {
    f4 aa = {a[0], a[1], a[2], a[3]}, bb = {b[0], b[1], b[2], b[3]};
    f4 r = smin4(aa, bb, k);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
float rmo_smin1(float a, float b, float k) { return smin1(a, b, k); }
float rmo_smax1(float a, float b, float k) { return smax1(a, b, k); }
float rmo_sq(float x) { return sq1(x); }
void rmo_sq3(const float x[3], float out[3]) { out[0] = sq1(x[0]); out[1] = sq1(x[1]); out[2] = sq1(x[2]); }
float rmo_luminance(const float c[3]) { return luminance3(v3(c[0], c[1], c[2])); }
/* shader:80-82  getnormal(p, s=0) = normalize(getnormalraw(p, s)); raw receives the un-normalised differences */
void rmo_getnormal(const rmo_uniforms *u, const float p[3], float out[3], float raw[3]) {
    scene_ctx c;
    scene_ctx_init(&c, u);
    f3 r = getnormalraw(&c, v3(p[0], p[1], p[2]));
    f3 n = normalize3(r);
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
    if (raw) { raw[0] = r.x; raw[1] = r.y; raw[2] = r.z; }
}
void rmo_scene(const rmo_uniforms *u, const float p[3], float out[4]) {
    scene_ctx c;
    scene_ctx_init(&c, u);
    f4 r = scene(&c, v3(p[0], p[1], p[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void rmo_pixel_ray(const rmo_uniforms *u, int x, int y, int w, int h, float rd[3]) {
    scene_ctx c;
    scene_ctx_init(&c, u);
    f3 r = pixel_ray(&c, x, y, w, h);
    rd[0] = r.x; rd[1] = r.y; rd[2] = r.z;
}
int rmo_raymarch(const rmo_uniforms *u, const float ro[3], const float rd[3], float depth,
                 float *length, int *evals) {
    scene_ctx c;
    scene_ctx_init(&c, u);
    march_t r = raymarch(&c, v3(ro[0], ro[1], ro[2]), v3(rd[0], rd[1], rd[2]), depth, u->contact_threshold);
    if (length) *length = r.length;
    if (evals) *evals = r.evals;
    return r.hit;
}
