// rm_math.h -- device-side SDF arithmetic of the march, written for gfx950.
//
// Every expression keeps the operand order of the shader source it replaces and
// the file is compiled with -ffp-contract=off, so each scene evaluation rounds
// exactly as a strict binary32 evaluation of the reference would.  What is
// changed is only what cannot change a bit of the result:
//   * uniform-only sub-expressions arrive pre-evaluated in RmPrimDev,
//   * the march loops evaluate the distance lane of smin() alone (its rgb lanes
//     never feed .w, Assets/Raymarching.cginc:110-113),
//   * k = .5 is a power of two, so 0.5*(b-a)/k == (b-a) and k*h == 0.5*h,
//   * rotate() by the identity quaternion and a divide by a unit scale are
//     skipped; a divide by a (wave-uniform) scale uses its precomputed reciprocal plus one exact
//     fma residual correction, which returns the correctly rounded quotient (rm_div_by).
#pragma once
#include <hip/hip_runtime.h>
#include "rm_device.h"

#define RM_DI __device__ __forceinline__

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

RM_DI V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RM_DI float rm_dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Correctly rounded sqrt for x == 0 or 2^-96 <= x <= 2^96 (sums of squared scene
// coordinates are either exactly 0 or far above 2^-96) in 6 VALU ops, one of them
// transcendental: y = v_rsq_f32(x) is within 1 ulp, g = x*y is a root estimate, the fma
// gives its residual x - g*g exactly, and one Newton step g + r*(y/2) lands on the correctly
// rounded root.  Verified against (float)sqrt((double)x) for every binary32 in [2^-96, 2^96]
// and 0 on gfx950 (scripts/ubench/sqrt_exhaustive.hip; raw v_sqrt_f32 is off by one ulp for 15 %
// of inputs).  The clamp keeps 0 * y finite at x == 0.  The generic expansion costs 16 ops.
#ifndef RM_SQRT_V1
RM_DI float rm_sqrt(float x) {
    float y = fminf(__builtin_amdgcn_rsqf(x), 0x1p+100f);
    float g = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(r, h, g);
}
#else
// earlier 9-op form: v_sqrt_f32, exact fma residuals against the lower and upper neighbour,
// their sign bits added into the bit pattern (also exhaustively verified)
RM_DI float rm_sqrt(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    uint32_t sb = __float_as_uint(s);
    uint32_t dn = __builtin_elementwise_sub_sat(sb, 1u);          // 0 stays 0
    float h_dn = __builtin_fmaf(__uint_as_float(dn), s, -x);      // >= 0: the root rounds down to dn
    float h_up = __builtin_fmaf(__uint_as_float(sb + 1u), s, -x); // <  0: the root rounds up to sb+1
    return __uint_as_float(dn + (__float_as_uint(h_dn) >> 31) + (__float_as_uint(h_up) >> 31));
}
#endif
RM_DI float rm_len3(V3 a) { return rm_sqrt(rm_dot3(a, a)); }
RM_DI float rm_len2(float x, float y) { return rm_sqrt(x * x + y * y); }

// The same root with the x == 0 guard as an add instead of a clamp: t must hold RM_SQRT_TINY = 2^-126 in a
// vector register.  x + t == x for every x >= 2^-96 (t is below a quarter ulp of x, round to nearest), so
// the estimate and with it the result are those of rm_sqrt; at x == 0 the estimate is the finite
// rsq(2^-126) and the result 0.  An add of two VGPRs issues in ~2.5 cycles, a min with a literal in ~4.2;
// the primitives run six roots per scene evaluation.
#define RM_SQRT_TINY 0x1p-126f
#ifndef RM_SQRT_V1
RM_DI float rm_sqrt_t(float x, float t) {
    float y = __builtin_amdgcn_rsqf(x + t);
    float g = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(r, h, g);
}
#else
RM_DI float rm_sqrt_t(float x, float) { return rm_sqrt(x); }
#endif
RM_DI float rm_len3(V3 a, float t) { return rm_sqrt_t(rm_dot3(a, a), t); }
RM_DI float rm_len2(float x, float y, float t) { return rm_sqrt_t(x * x + y * y, t); }
RM_DI float rm_lerp(float a, float b, float t) { return a + t * (b - a); }
RM_DI V3 rm_lerp3(V3 a, V3 b, float t) { return mk3(rm_lerp(a.x, b.x, t), rm_lerp(a.y, b.y, t), rm_lerp(a.z, b.z, t)); }
RM_DI float rm_clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
RM_DI V3 rm_normalize(V3 a) { float r = 1.0f / sqrtf(rm_dot3(a, a)); return mk3(a.x * r, a.y * r, a.z * r); }   // once per pixel: generic sqrt

// Scene classes whose march loops may round differently from a strict binary32 evaluation (the opt-in tolerance
// kernel, rm_fast.h) say so here; the loops then fuse their own multiply-adds too.
template <class S> struct rm_is_fast { static constexpr bool value = false; };
// A scene type can promise a plain launch shape: fragment pass, whole frames, no source image, no depth texture, all three
// output planes, frame size with exact reciprocals -- either one frame with the view in the kernel arguments (1) or a
// batch of poses with the views in device memory (2); or one frame as (1) but WITH source image and depth texture, which
// is what Graphics.Blit(src, dest, mat) of the reference passes (3); or one frame as (1) whose shadow term is a lookup in the bound
// _ShadowMap instead of the soft-shadow march (4).  The per-tile code then drops the uniform branches (and the scalar
// registers their conditions occupy) that the other launch shapes need.  0 = any shape.
template <class S> struct rm_shape { static constexpr int value = 0; };
// A scene type whose root is smin(left(p), right(p)) with 1-Lipschitz operands may offer the two halves separately: the
// soft-shadow march then skips the left half while it provably lies .5 above the right one (rm_getshadow).
template <class S> struct rm_pair_skip { static constexpr bool value = false; };

// scale handling chosen per scene on the host
enum { RM_SCL_UNIT = 0, RM_SCL_RCP = 1, RM_SCL_DIV = 2 };

// x / s from r = RN(1/s):  q0 = x*r;  e = x - q0*s (exact, one fma);  q = q0 + e*r (one fma).
// With a correctly rounded reciprocal this is the correctly rounded quotient (Markstein's
// correction step); checked against IEEE division for every binary32 x in [2^-60, 2^60] and 100
// divisors including all-ones and near-power-of-two significands (scripts/ubench/div_exhaustive.hip).
// 3 full-rate VALU ops instead of the ~12 (one quarter-rate) of a correctly rounded divide.  The host
// only selects it for normal scales with |exponent| <= 60 and a significand that is not all ones.
RM_DI float rm_div_by(float x, float s, float r) {
    float q0 = x * r;
    float e = __builtin_fmaf(-q0, s, x);
    return __builtin_fmaf(e, r, q0);
}

// p_i = rotate(p - pos, q) / scale        (cginc:336-354, rotate at :62-66)
template <bool ROT, int SCL, class PT>
RM_DI V3 rm_to_local(const PT &P, V3 p) {
    V3 v = mk3(p.x - P.px, p.y - P.py, p.z - P.pz);
    if (ROT) {
        V3 u = mk3(P.ux, P.uy, P.uz);
        float a = 2.0f * rm_dot3(u, v);
        V3 cr = mk3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
        v = mk3((u.x * a + v.x * P.c) + cr.x * P.w2,
                (u.y * a + v.y * P.c) + cr.y * P.w2,
                (u.z * a + v.z * P.c) + cr.z * P.w2);
    }
    if (SCL == RM_SCL_RCP) v = mk3(rm_div_by(v.x, P.sx, P.rx), rm_div_by(v.y, P.sy, P.ry), rm_div_by(v.z, P.sz, P.rz));
    if (SCL == RM_SCL_DIV) v = mk3(v.x / P.sx, v.y / P.sy, v.z / P.sz);
    return v;
}

// cginc:169-171; a0 = r*.5
template <class PT> RM_DI float rm_sphere(const PT &P, V3 p, float t) { return rm_len3(p, t) - P.a0; }
// cginc:173-177; a0..a2 = b*.5
template <class PT> RM_DI float rm_box(const PT &P, V3 p, float t) {
    float dx = fabsf(p.x) - P.a0, dy = fabsf(p.y) - P.a1, dz = fabsf(p.z) - P.a2;
    V3 m = mk3(fmaxf(dx, 0.0f), fmaxf(dy, 0.0f), fmaxf(dz, 0.0f));
    return rm_len3(m, t) + fminf(fmaxf(dx, fmaxf(dy, dz)), 0.0f);
}
// cginc:184-187; a0 = le, a1 = r1, a2 = r2
template <class PT> RM_DI float rm_link(const PT &P, V3 p, float t) {
    float qy = fmaxf(fabsf(p.y) - P.a0, 0.0f);
    return rm_len2(rm_len2(p.x, qy, t) - P.a1, p.z, t) - P.a2;
}
// cginc:189-192; a0 = h (radius), a1 = r (half height)
template <class PT> RM_DI float rm_cylinder(const PT &P, V3 p, float t) {
    float dx = fabsf(rm_len2(p.x, p.z, t)) - P.a0;
    float dy = fabsf(p.y) - P.a1;
    return fminf(fmaxf(dx, dy), 0.0f) + rm_len2(fmaxf(dx, 0.0f), fmaxf(dy, 0.0f), t);
}
// cginc:179-182; a0 = t.x, a1 = t.y
template <class PT> RM_DI float rm_torus(const PT &P, V3 p, float t) {
    return rm_len2(rm_len2(p.x, p.y, t) - P.a0, p.z, t) - P.a1;
}

// distance lane of smin(float4 a, float4 b, .5)  (cginc:110-113)
RM_DI float rm_smin_h(float a, float b) { return rm_clamp01(0.5f + (b - a)); }
// lerp(b,a,h) = b + h*(a-b) is evaluated as b - h*(b-a): a-b == -(b-a) and h*(-e) == -(h*e)
// hold exactly in binary floating point, so one subtraction serves both h and the lerp
RM_DI float rm_smin_w(float a, float b) {
    float e = b - a;
    float h = rm_clamp01(0.5f + e);
    return (b - h * e) - (0.5f * h) * (1.0f - h);
}
// full smin: rgb through blend() (cginc:101-103)
RM_DI V4 rm_smin4(V4 a, V4 b) {
    float h = rm_smin_h(a.w, b.w);
    V3 c1 = mk3(b.x, b.y, b.z), c2 = mk3(a.x, a.y, a.z);
    V3 n = rm_normalize(rm_lerp3(c1, c2, h));
    float l = rm_lerp(rm_len3(c1), rm_len3(c2), h);
    V4 r;
    r.x = n.x * l; r.y = n.y * l; r.z = n.z * l;
    r.w = rm_lerp(b.w, a.w, h) - (0.5f * h) * (1.0f - h);
    return r;
}

// smax(float4 a, float4 b, .5)  (cginc:120-123): plain lerp of rgb; the k*h*(1-h) term keeps the
// reference's sign
RM_DI V4 rm_smax4(V4 a, V4 b) {
    float h = rm_clamp01(0.5f + (a.w - b.w));
    V3 c = rm_lerp3(mk3(b.x, b.y, b.z), mk3(a.x, a.y, a.z), h);
    V4 r;
    r.x = c.x; r.y = c.y; r.z = c.z;
    r.w = -rm_lerp(-b.w, -a.w, h) - (0.5f * h) * (1.0f - h);
    return r;
}

template <bool ROT, int SCL, class PT>
RM_DI float rm_prim_typed(const PT &P, int type, V3 p, float t) {
    V3 l = rm_to_local<ROT, SCL>(P, p);
    float d;
    switch (type) {
    case 0:  d = rm_sphere(P, l, t); break;
    case 1:  d = rm_box(P, l, t); break;
    case 2:  d = rm_link(P, l, t); break;
    case 3:  d = rm_cylinder(P, l, t); break;
    default: d = rm_torus(P, l, t); break;
    }
    return SCL == RM_SCL_UNIT ? d : d * P.ms;
}

// A wave-uniform value moved into a VGPR once.  On gfx950 a VALU op with an SGPR (or
// literal) source issues at half the rate of the same op on VGPRs, so constants that the
// march loop touches every iteration are kept in vector registers.
RM_DI float rm_vgpr(float s) {
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// vector-register copy of one prepared primitive (only the fields the variant reads)
struct PrimV {
    float px, py, pz, ux, uy, uz, c, w2, sx, sy, sz, rx, ry, rz, ms, a0, a1, a2;
};
template <bool ROT, int SCL>
RM_DI PrimV rm_prim_to_vgpr(const RmPrimDev &P) {
    PrimV v;
    v.px = rm_vgpr(P.px); v.py = rm_vgpr(P.py); v.pz = rm_vgpr(P.pz);
    v.a0 = rm_vgpr(P.a0); v.a1 = rm_vgpr(P.a1); v.a2 = rm_vgpr(P.a2);
    if (ROT) { v.ux = rm_vgpr(P.ux); v.uy = rm_vgpr(P.uy); v.uz = rm_vgpr(P.uz); v.c = rm_vgpr(P.c); v.w2 = rm_vgpr(P.w2); }
    else     { v.ux = v.uy = v.uz = v.c = v.w2 = 0.0f; }
    if (SCL != RM_SCL_UNIT) { v.sx = rm_vgpr(P.sx); v.sy = rm_vgpr(P.sy); v.sz = rm_vgpr(P.sz); v.ms = rm_vgpr(P.ms); }
    else                    { v.sx = v.sy = v.sz = v.ms = 1.0f; }
    if (SCL == RM_SCL_RCP) { v.rx = rm_vgpr(P.rx); v.ry = rm_vgpr(P.ry); v.rz = rm_vgpr(P.rz); }
    else                   { v.rx = v.ry = v.rz = 1.0f; }
    return v;
}

// The reference's fixed scene: sphere, box, link, cylinder in slots 0..3
// (cginc:351-354) combined as smin(smin(o0,o1), smin(o2,o3)) (cginc:363).
template <bool ROT, int SCL>
struct Scene4 {
    const RmParams &U;
    PrimV P0, P1, P2, P3;
    float tiny;
    RM_DI explicit Scene4(const RmParams &u) : U(u) {
        tiny = rm_vgpr(RM_SQRT_TINY);
        P0 = rm_prim_to_vgpr<ROT, SCL>(u.prims4[0]);
        P1 = rm_prim_to_vgpr<ROT, SCL>(u.prims4[1]);
        P2 = rm_prim_to_vgpr<ROT, SCL>(u.prims4[2]);
        P3 = rm_prim_to_vgpr<ROT, SCL>(u.prims4[3]);
    }
    RM_DI float w(V3 p) const {
        float o0 = rm_prim_typed<ROT, SCL>(P0, 0, p, tiny);
        float o1 = rm_prim_typed<ROT, SCL>(P1, 1, p, tiny);
        float o2 = rm_prim_typed<ROT, SCL>(P2, 2, p, tiny);
        float o3 = rm_prim_typed<ROT, SCL>(P3, 3, p, tiny);
        return rm_smin_w(rm_smin_w(o0, o1), rm_smin_w(o2, o3));
    }
    RM_DI float wm(V3 p) const { return w(p); }      // the evaluation the two march loops use (a kernel may relax it)
    // the two operands of the root smin (cginc:363): w(p) == rm_smin_w(left(p), right(p)), operation for operation
    RM_DI float left(V3 p) const { return rm_smin_w(rm_prim_typed<ROT, SCL>(P0, 0, p, tiny), rm_prim_typed<ROT, SCL>(P1, 1, p, tiny)); }
    RM_DI float right(V3 p) const { return rm_smin_w(rm_prim_typed<ROT, SCL>(P2, 2, p, tiny), rm_prim_typed<ROT, SCL>(P3, 3, p, tiny)); }
    RM_DI bool lipschitz() const { return (U.scene_flags & RM_SF_LIPSCHITZ) != 0; }
    RM_DI V4 full(V3 p) const {
        V4 o[4];
        o[0].w = rm_prim_typed<ROT, SCL>(P0, 0, p, tiny);
        o[1].w = rm_prim_typed<ROT, SCL>(P1, 1, p, tiny);
        o[2].w = rm_prim_typed<ROT, SCL>(P2, 2, p, tiny);
        o[3].w = rm_prim_typed<ROT, SCL>(P3, 3, p, tiny);
#pragma unroll
        for (int i = 0; i < 4; i++) { o[i].x = U.prims4[i].cr; o[i].y = U.prims4[i].cg; o[i].z = U.prims4[i].cb; }
        return rm_smin4(rm_smin4(o[0], o[1]), rm_smin4(o[2], o[3]));
    }
};

template <bool ROT, int SCL, int SHAPE>
struct Scene4Plain : Scene4<ROT, SCL> {
    RM_DI explicit Scene4Plain(const RmParams &u) : Scene4<ROT, SCL>(u) {}
};
template <bool ROT, int SCL, int SHAPE> struct rm_shape<Scene4Plain<ROT, SCL, SHAPE>> { static constexpr int value = SHAPE; };
// (scenes without per-primitive rotation only: the rotated variants already fill their 256 registers, and the second copy of
// the right pair's code that the skipping loop needs made them spill inside the march; the reciprocal-exact scale variant,
// compiled for three waves per SIMD, spilled 32 bytes with it and gained nothing)
#ifndef RM_NO_PAIR_SKIP
template <int SCL> struct rm_pair_skip<Scene4<false, SCL>> { static constexpr bool value = SCL != RM_SCL_RCP; };
template <int SCL, int SHAPE> struct rm_pair_skip<Scene4Plain<false, SCL, SHAPE>> { static constexpr bool value = SCL != RM_SCL_RCP; };
#endif

// N-primitive scene from a table in global memory.  Every lane walks the table
// with the same index, so the loads are wave-uniform and go through the scalar
// cache.  The balanced pairwise tree "in index order" (level by level, an odd
// last element carried up) is evaluated in one pass with a binary-counter stack:
// after primitive i, stack[l] holds the finished subtree of 2^l leaves iff bit l
// of (i+1) is set; leftovers are merged low level first.
typedef unsigned rm_u4 __attribute__((ext_vector_type(4)));
typedef const rm_u4 __attribute__((address_space(3))) *RmLdsPrimPtr;      // the table as 16-byte words in LDS

// fetches record i either from the LDS copy (ds_read broadcast into VGPRs) or from global
// memory (wave-uniform address -> scalar cache)
template <bool LDS> struct PrimTable;
template <> struct PrimTable<true> {
    RmLdsPrimPtr T;
    RM_DI RmPrimDev get(int i) const {
        union { RmPrimDev p; rm_u4 v[sizeof(RmPrimDev) / 16]; } u;
#pragma unroll
        for (int j = 0; j < (int)(sizeof(RmPrimDev) / 16); j++) u.v[j] = T[i * (int)(sizeof(RmPrimDev) / 16) + j];   // ds_read_b128, same address in every lane
        // every lane holds the same record: say so for the word that steers the control flow, so that the choice of the
        // primitive's code is a scalar branch instead of three vector compares with their exec-mask bookkeeping
        u.p.type = __builtin_amdgcn_readfirstlane(u.p.type);
        return u.p;
    }
};
template <> struct PrimTable<false> {
    const RmPrimDev *__restrict__ T;
    RM_DI const RmPrimDev &get(int i) const { return T[i]; }
};

#define RM_ITEM_LEVELS 9      // binary-counter levels over groups of four leaves: 4 * 2^8 = 1024 primitives

// N-primitive scene.  The balanced pairwise tree "in index order" (level by level, an odd
// last element carried up unchanged) is evaluated in one pass over the table:
//   * four consecutive leaves form one complete level-2 subtree smin(smin(a,b), smin(c,d)),
//     evaluated straight-line (a trailing group of 1..3 leaves gives a, smin(a,b) or
//     smin(smin(a,b), c) -- exactly what the carry rule makes of it);
//   * the level-2 items are merged with a binary-counter stack: after item j, st[l] holds the
//     finished subtree of 2^l items iff bit l of (j+1) is set; leftovers are merged low level
//     first at the end, which reproduces the carry rule for any count.
// All control flow is wave-uniform (every lane walks the same table), so it runs on the scalar
// unit; grouping by four keeps that scalar work to a few instructions per primitive.
template <bool ROT, int SCL, bool LDS>
struct SceneN {
    PrimTable<LDS> tab;
    int n;
    bool csg;              // some primitive carries an op bit
    uint32_t tail_ops;
    float tiny;
    RM_DI SceneN(PrimTable<LDS> t, int n_, bool csg_, uint32_t tail_ops_) : tab(t), n(n_), csg(csg_), tail_ops(tail_ops_) { tiny = rm_vgpr(RM_SQRT_TINY); }

    // leaf value plus what the tree needs from the record: osgn (the op of the node this leaf
    // opens as leftmost leaf of a right operand) and the type word with the cascade bits
    template <bool CSG>
    RM_DI float leaf_w(int i, V3 p, float &osgn, int &tw) const {
        const auto P = tab.get(i);
        float d = rm_prim_typed<ROT, SCL>(P, P.type & RM_TYPE_MASK, p, tiny);
        if (CSG) { d = d * P.nsgn; osgn = P.osgn; tw = P.type; }       // x * -1 == -x exactly
        return d;
    }
    template <bool CSG>
    RM_DI V4 leaf4(int i, V3 p, float &osgn, int &tw) const {
        const auto P = tab.get(i);
        V4 v;
        v.w = rm_prim_typed<ROT, SCL>(P, P.type & RM_TYPE_MASK, p, tiny);
        if (CSG) { v.w = v.w * P.nsgn; osgn = P.osgn; tw = P.type; }
        v.x = P.cr; v.y = P.cg; v.z = P.cb;
        return v;
    }
    // node of the tree: s = +1 smin (cginc:110-113), s = -1 smax (cginc:120-123); s is wave-uniform
    template <bool CSG> RM_DI static float comb(float a, float b, float s) {
        if (!CSG) return rm_smin_w(a, b);
        float e = b - a;
        float h = rm_clamp01(0.5f + e * s);          // smax: 0.5 + (a - b), and a - b == -(b - a) exactly
        return (b - h * e) - (0.5f * h) * (1.0f - h);
    }
    template <bool CSG> RM_DI static V4 comb(V4 a, V4 b, float s) {
        if (!CSG || s > 0.0f) return rm_smin4(a, b);
        return rm_smax4(a, b);
    }

    template <class T, bool CSG, class LEAF>
    RM_DI T tree(LEAF leaf) const {
        T st[RM_ITEM_LEVELS];
        const int groups = n >> 2, rem = n & 3;
        for (int g = 0; g < groups; g++) {
            const int i = g << 2;
            float s0 = 1.0f, sb = 1.0f, sc = 1.0f, sd = 1.0f;
            int tw = 0, t1 = 0;
            T a = leaf(i, s0, tw), b = leaf(i + 1, sb, t1), c = leaf(i + 2, sc, t1), d = leaf(i + 3, sd, t1);
            T v = comb<CSG>(comb<CSG>(a, b, sb), comb<CSG>(c, d, sd), sc);
            const int merges = __builtin_ctz(~(unsigned)g);          // trailing ones of g = finished lower levels
#pragma unroll
            for (int l = 0; l < RM_ITEM_LEVELS - 1; l++)
                if (l < merges) v = comb<CSG>(st[l], v, (CSG && ((tw >> (RM_CASCADE_SHIFT + l)) & 1)) ? -1.0f : 1.0f);
#pragma unroll
            for (int l = 0; l < RM_ITEM_LEVELS; l++) if (l == merges) st[l] = v;       // (a switch on the uniform `merges` was measured: slower)
        }
        int items = groups;
        T acc = T();
        bool have = false;
        if (rem) {
            const int i = groups << 2;
            float s0 = 1.0f, sb = 1.0f, sc = 1.0f;
            int t1 = 0;
            acc = leaf(i, s0, t1);
            if (rem >= 2) { T b = leaf(i + 1, sb, t1); acc = comb<CSG>(acc, b, sb); }
            if (rem == 3) { T c = leaf(i + 2, sc, t1); acc = comb<CSG>(acc, c, sc); }
            have = true;
        }
#pragma unroll
        for (int l = 0; l < RM_ITEM_LEVELS; l++) {
            if (items & (1 << l)) {
                acc = have ? comb<CSG>(st[l], acc, (CSG && ((tail_ops >> l) & 1u)) ? -1.0f : 1.0f) : st[l];
                have = true;
            }
        }
        return acc;
    }
    RM_DI float w(V3 p) const {
        if (csg) return tree<float, true>([&](int i, float &s, int &tw) { return leaf_w<true>(i, p, s, tw); });
        return tree<float, false>([&](int i, float &s, int &tw) { return leaf_w<false>(i, p, s, tw); });
    }
    RM_DI float wm(V3 p) const { return w(p); }
    RM_DI V4 full(V3 p) const {
        if (csg) return tree<V4, true>([&](int i, float &s, int &tw) { return leaf4<true>(i, p, s, tw); });
        return tree<V4, false>([&](int i, float &s, int &tw) { return leaf4<false>(i, p, s, tw); });
    }
};

// ---- fractal scene() bodies -------------------------------------------------------------
// rotate(float3 v, float3 a) of cginc:51-60 with the cos/sin of the uniform angles taken on the
// host; the constant 0/1 matrix entries are folded (a zero result may differ in sign, which the
// following s*x - c discards).
struct EulerV { float cx, sx, cy, sy, cz, sz; };
RM_DI V3 rm_rotate_euler(const EulerV &e, V3 v) {
    const float ay = e.cx * v.y - e.sx * v.z, az = e.sx * v.y + e.cx * v.z;          // mx
    const float bx = e.cy * v.x + e.sy * az,  bz = e.cy * az - e.sy * v.x;           // my
    return mk3(e.cz * bx - e.sz * ay, e.sz * bx + e.cz * ay, bz);                    // mz
}

// ---- canonical transcendental functions -----------------------------------------------------
// HLSL does not bit-define sin/cos/acos/asin/atan2/pow/log, so the fractal bodies that use them
// (cginc:194-262) have no bit-defined result in the reference.  They are fixed as float -> float
// functions evaluated in IEEE double with +,-,*,/ and sqrt in exactly this operation order (no
// fused multiply-add), rounded to float once: every IEEE machine produces the same bits, and each
// result is the correctly rounded float of the true function in all sampled cases.
#define RM_PIO2_HI 1.57079632673412561417e+00   /* first 33 bits of pi/2 */
#define RM_PIO2_LO 6.07710050650619224932e-11   /* pi/2 - RM_PIO2_HI */
#define RM_PI      3.14159265358979311600e+00
#define RM_PIO2    1.57079632679489655800e+00
#define RM_PIO4    7.85398163397448278999e-01
#define RM_LN2     6.93147180559945286227e-01
#define RM_LN2_HI  6.93147180369123816490e-01   /* first 32 bits of ln 2 */
#define RM_LN2_LO  1.90821492927058770002e-10   /* ln 2 - RM_LN2_HI */

RM_DI double rm_c_sin_poly(double r) {          // Taylor series on [-pi/4, pi/4], Horner on r*r
    const double q = r * r;
    double p = 1.0 / 355687428096000.0;
    p = p * q - 1.0 / 1307674368000.0;
    p = p * q + 1.0 / 6227020800.0;
    p = p * q - 1.0 / 39916800.0;
    p = p * q + 1.0 / 362880.0;
    p = p * q - 1.0 / 5040.0;
    p = p * q + 1.0 / 120.0;
    p = p * q - 1.0 / 6.0;
    return r + r * (q * p);
}
RM_DI double rm_c_cos_poly(double r) {
    const double q = r * r;
    double p = 1.0 / 20922789888000.0;
    p = p * q - 1.0 / 87178291200.0;
    p = p * q + 1.0 / 479001600.0;
    p = p * q - 1.0 / 3628800.0;
    p = p * q + 1.0 / 40320.0;
    p = p * q - 1.0 / 720.0;
    p = p * q + 1.0 / 24.0;
    p = p * q - 0.5;
    return 1.0 + q * p;
}
// sin and cos of one angle share the reduction x = k*pi/2 + r and the two series
RM_DI void rm_c_sincos(float xf, float &sn, float &cs) {
    const double x = (double)xf;
    const double k = __builtin_floor(x * (1.0 / RM_PIO2) + 0.5);
    const double r = (x - k * RM_PIO2_HI) - k * RM_PIO2_LO;
    const int n = (int)(k - 4.0 * __builtin_floor(k * 0.25));
    const double sp = rm_c_sin_poly(r), cp = rm_c_cos_poly(r);
    const double vs = (n & 1) ? cp : sp, vc = (n & 1) ? sp : cp;
    sn = (float)((n & 2) ? -vs : vs);
    cs = (float)(((n + 1) & 2) ? -vc : vc);
}
RM_DI double rm_c_atan01(double t) {            // atan on [0, 1]
    double base = 0.0, u = t;
    if (t > 0.41421356237309503) { u = (t - 1.0) / (t + 1.0); base = RM_PIO4; }
    const double q = u * u;
    double p = 1.0 / 43.0;
#pragma unroll
    for (int n = 41; n >= 1; n -= 2) p = 1.0 / (double)n - q * p;
    return base + u * p;
}
RM_DI double rm_c_atan2d(double y, double x) {
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    double a;
    if (ay > ax) a = RM_PIO2 - rm_c_atan01(ax / ay);
    else if (ax > 0.0) a = rm_c_atan01(ay / ax);
    else a = ay / ax;                           // 0/0 and NaN stay NaN
    if (x < 0.0) a = RM_PI - a;
    return (y < 0.0) ? -a : a;
}
RM_DI float rm_c_atan2(float y, float x) { return (float)rm_c_atan2d((double)y, (double)x); }
RM_DI float rm_c_acos(float vf) {
    const double v = (double)vf;
    double t = 1.0 - v * v;
    if (!(t > 0.0)) t = 0.0;
    return (float)rm_c_atan2d(__builtin_sqrt(t), v);
}
RM_DI float rm_c_asin(float vf) {
    const double v = (double)vf;
    double t = 1.0 - v * v;
    if (!(t > 0.0)) t = 0.0;
    return (float)rm_c_atan2d(v, __builtin_sqrt(t));
}
RM_DI double rm_c_logd(double x) {              // finite x > 0
    int e;
    double m = __builtin_frexp(x, &e) * 2.0;
    e -= 1;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), q = s * s;
    double p = 1.0 / 25.0;
#pragma unroll
    for (int n = 23; n >= 1; n -= 2) p = 1.0 / (double)n + q * p;
    return (double)e * RM_LN2 + 2.0 * s * p;
}
RM_DI double rm_c_expd(double y) {
    if (!(y < 700.0)) return y > 0.0 ? (double)__builtin_huge_valf() : y;
    if (y < -700.0) return 0.0;
    const double k = __builtin_floor(y * (1.0 / RM_LN2) + 0.5);
    const double f = (y - k * RM_LN2_HI) - k * RM_LN2_LO;
    double p = 1.0 / 87178291200.0;
    p = 1.0 / 6227020800.0 + f * p;
    p = 1.0 / 479001600.0 + f * p;
    p = 1.0 / 39916800.0 + f * p;
    p = 1.0 / 3628800.0 + f * p;
    p = 1.0 / 362880.0 + f * p;
    p = 1.0 / 40320.0 + f * p;
    p = 1.0 / 5040.0 + f * p;
    p = 1.0 / 720.0 + f * p;
    p = 1.0 / 120.0 + f * p;
    p = 1.0 / 24.0 + f * p;
    p = 1.0 / 6.0 + f * p;
    p = 1.0 / 2.0 + f * p;
    p = 1.0 + f * p;
    p = 1.0 + f * p;
    return __builtin_ldexp(p, (int)k);
}
RM_DI float rm_c_log(float xf) {
    if (!(xf > 0.0f)) return xf == 0.0f ? -__builtin_huge_valf() : __builtin_nanf("");
    if (__builtin_isinf(xf)) return xf;
    return (float)rm_c_logd((double)xf);
}
// pow(x, e) for x >= 0 from ln x (so that two powers of one x share it)
RM_DI float rm_c_pow_ln(float xf, double lnx, float ef) {
    if (!(xf > 0.0f)) {
        if (xf == 0.0f) return ef > 0.0f ? 0.0f : (ef == 0.0f ? 1.0f : __builtin_huge_valf());
        return __builtin_nanf("");
    }
    if (__builtin_isinf(xf)) return ef > 0.0f ? xf : (ef == 0.0f ? 1.0f : 0.0f);
    return (float)rm_c_expd((double)ef * lnx);
}

// Fractal scene: body(rotate(p - pos, q) / scale) with .w scaled by min(scale), pos/q/scale from
// slot 0 of the three matrices.  The orbit traps (rgb) are minima of lengths; sqrt is monotonic
// and correctly rounded here, so the minimum is taken over the squared lengths and rooted once,
// and the marches (which read only .w) skip the traps altogether.
struct SceneF {
    const RmParams &U;
    PrimV P0;
    EulerV E, Et, Es;      // _FractalRotationX/Y/Z; composite body: the bulb's and the sponge's rotation of p (cginc:349)
    float three, two, scale, scale_m1;
    bool rot, unit, rcp;
    RM_DI explicit SceneF(const RmParams &u) : U(u) {
        P0 = rm_prim_to_vgpr<true, RM_SCL_RCP>(u.prims4[0]);
        E.cx = rm_vgpr(u.f_rot[0]); E.sx = rm_vgpr(u.f_rot[1]); E.cy = rm_vgpr(u.f_rot[2]);
        E.sy = rm_vgpr(u.f_rot[3]); E.cz = rm_vgpr(u.f_rot[4]); E.sz = rm_vgpr(u.f_rot[5]);
        Et.cx = u.f_rot_time[0]; Et.sx = u.f_rot_time[1]; Et.cy = u.f_rot_time[2]; Et.sy = u.f_rot_time[3]; Et.cz = u.f_rot_time[4]; Et.sz = u.f_rot_time[5];
        Es.cx = u.f_rot_sponge[0]; Es.sx = u.f_rot_sponge[1]; Es.cy = u.f_rot_sponge[2]; Es.sy = u.f_rot_sponge[3]; Es.cz = u.f_rot_sponge[4]; Es.sz = u.f_rot_sponge[5];
        three = rm_vgpr(3.0f); two = rm_vgpr(2.0f);
        scale = rm_vgpr(u.f_scale); scale_m1 = rm_vgpr(u.f_scale_m1);
        rot = !(u.scene_flags & RM_SF_IDENT_ROT);
        unit = (u.scene_flags & RM_SF_UNIT_SCALE) != 0;
        rcp = (u.scene_flags & RM_SF_RCP) != 0;
    }
    RM_DI V3 local(V3 p) const {                      // all three branches are wave-uniform
        if (rot) { if (unit) return rm_to_local<true, RM_SCL_UNIT>(P0, p); if (rcp) return rm_to_local<true, RM_SCL_RCP>(P0, p); return rm_to_local<true, RM_SCL_DIV>(P0, p); }
        if (unit) return rm_to_local<false, RM_SCL_UNIT>(P0, p);
        if (rcp) return rm_to_local<false, RM_SCL_RCP>(P0, p);
        return rm_to_local<false, RM_SCL_DIV>(P0, p);
    }
    // cginc:295-313
    template <bool TRAPS> RM_DI V4 menger(V3 p) const {
        float x = p.x, y = p.y, z = p.z;
        float t0 = 2500.0f, t1 = 2500.0f, t2 = 2500.0f;          // 50^2
#pragma nounroll
        for (int n = 0; n < 12; n++) {
            x = fabsf(x); y = fabsf(y); z = fabsf(z);
            float hi = fmaxf(x, y), lo = fminf(x, y); x = hi; y = lo;      // if (x < y) swap
            hi = fmaxf(y, z); lo = fminf(y, z); y = hi; z = lo;
            hi = fmaxf(x, y); lo = fminf(x, y); x = hi; y = lo;
            V3 q = rm_rotate_euler(E, mk3(x, y, z));
            x = q.x * three - two; y = q.y * three - two; z = q.z * three - two;
            if (z < -1.0f) z += 2.0f;
            if (TRAPS) {
                const float yy = y - 0.5f, zz = z - 0.5f;
                t0 = fminf(t0, x * x + y * y + z * z);
                t1 = fminf(t1, x * x + yy * yy + z * z);
                t2 = fminf(t2, x * x + y * y + zz * zz);
            }
        }
        V4 r;
        r.w = (rm_sqrt(x * x + y * y + z * z) - 1.5f) * U.f_pow3;
        if (TRAPS) { r.x = rm_sqrt(t0); r.y = rm_sqrt(t1); r.z = rm_sqrt(t2); } else { r.x = r.y = r.z = 0.0f; }
        return r;
    }
    // cginc:264-293
    template <bool TRAPS> RM_DI V4 sierpinski(V3 p) const {
        float x = p.x, y = p.y, z = p.z;
        float r2 = x * x + y * y + z * z;
        float t0 = 400.0f, t1 = 400.0f, t2 = 400.0f;             // bailout^2
        float pw = U.f_pows[0];
        for (int i = 0; i < 15; i++) {                           // i is the same in every lane still looping
            if (!(r2 < 20.0f)) break;
            if (x + y < 0.0f) { const float x1 = -y; y = -x; x = x1; }
            if (x + z < 0.0f) { const float x1 = -z; z = -x; x = x1; }
            if (y + z < 0.0f) { const float y1 = -z; z = -y; y = y1; }
            V3 c = rm_rotate_euler(E, mk3(x, y, z));
            x = scale * c.x - scale_m1; y = scale * c.y - scale_m1; z = scale * c.z - scale_m1;
            r2 = x * x + y * y + z * z;
            pw = U.f_pows[i + 1];
            if (TRAPS) {
                const float xx = x - 1.0f, yy = y - 1.0f, zz = z - 1.0f;
                t0 = fminf(t0, xx * xx + y * y + z * z);
                t1 = fminf(t1, x * x + yy * yy + z * z);
                t2 = fminf(t2, x * x + y * y + zz * zz);
            }
        }
        V4 r;
        r.w = (rm_sqrt(r2) - 2.0f) * pw;
        if (TRAPS) { r.x = rm_sqrt(t0); r.y = rm_sqrt(t1); r.z = rm_sqrt(t2); } else { r.x = r.y = r.z = 0.0f; }
        return r;
    }
    // cginc:194-227 (BROT false) and cginc:229-262 (BROT true); iters = 12, bailout = 10
    template <bool TRAPS, bool BROT> RM_DI V4 bulb(V3 p, float e) const {
        const float em1 = e - 1.0f;
        V3 z = p;
        float dr = 1.0f, r = 0.0f;
        float t0 = 100.0f, t1 = 100.0f, t2 = 100.0f;             // bailout^2
#pragma nounroll
        for (int i = 0; i < 12; i++) {
            const float r2 = z.x * z.x + z.y * z.y + z.z * z.z;
            r = rm_sqrt(r2);
            if (TRAPS) {
                t0 = fminf(t0, r2);
                if (BROT) {
                    const float a = z.x - 1.0f, b = z.x - 2.0f;
                    t1 = fminf(t1, a * a + z.y * z.y + z.z * z.z);
                    t2 = fminf(t2, b * b + z.y * z.y + z.z * z.z);
                } else {
                    const float a = z.z - 1.0f, b = z.z - 2.0f;
                    t1 = fminf(t1, z.x * z.x + z.y * z.y + a * a);
                    t2 = fminf(t2, z.x * z.x + z.y * z.y + b * b);
                }
            }
            if (r > 10.0f) break;
            const float ratio = z.z / r;
            float theta, phi;
            if (BROT) { theta = rm_c_atan2(z.y, z.x); phi = rm_c_asin(ratio); }
            else      { theta = rm_c_acos(ratio);     phi = rm_c_atan2(z.y, z.x); }
            const bool plain = r > 0.0f && !__builtin_isinf(r);
            const double lnr = plain ? rm_c_logd((double)r) : 0.0;
            dr = rm_c_pow_ln(r, lnr, em1) * e * dr + 1.0f;
            const float zr = rm_c_pow_ln(r, lnr, e);
            theta = theta * e;
            phi = phi * e;
            float st, ct, sp, cp;
            rm_c_sincos(theta, st, ct);
            rm_c_sincos(phi, sp, cp);
            if (BROT) z = mk3(zr * (ct * cp), zr * (st * cp), zr * (-sp));
            else      z = mk3(zr * (st * cp), zr * (sp * st), zr * ct);
            z = mk3(z.x + p.x, z.y + p.y, z.z + p.z);
        }
        V4 o;
        o.w = 0.5f * rm_c_log(r) * r / dr;
        if (TRAPS) { o.x = rm_sqrt(t0); o.y = rm_sqrt(t1); o.z = rm_sqrt(t2); } else { o.x = o.y = o.z = 0.0f; }
        return o;
    }
    // cginc:125-149: rand / noise with the canonical sin (double series) and frac(x) = x - floor(x)
    RM_DI static float rand2(float x, float y) {
        float sn, cs;
        rm_c_sincos(x * 12.9898f + y * 78.233f, sn, cs);
        const float v = sn * 43758.5453f;
        return v - floorf(v);
    }
    RM_DI static float noise2(float x, float y) {
        const float ix = floorf(x), iy = floorf(y);
        const float fx = x - ix, fy = y - iy;
        const float a = rand2(ix, iy), b = rand2(ix + 1.0f, iy + 0.0f), c = rand2(ix + 0.0f, iy + 1.0f), d = rand2(ix + 1.0f, iy + 1.0f);
        const float ux = fx * fx * (3.0f - 2.0f * fx), uy = fy * fy * (3.0f - 2.0f * fy);
        return rm_lerp(a, b, ux) + (c - a) * uy * (1.0f - ux) + (d - b) * ux * uy;
    }
    // cginc:315-330 as a body: t = terrain(p.xz); float4(t.rgb, p.y - t.w)
    RM_DI V4 terrain(V3 p) const {
        float px = p.x * 0.003f / 250.0f, py = p.z * 0.003f / 250.0f;
        float a = 0.0f, b = 1.0f, dx = 0.0f, dy = 0.0f;
#pragma nounroll
        for (int i = 0; i < 3; i++) {
            const float n = noise2(px, py);
            dx += n; dy += 0.0f;
            a += b * n / (1.0f + (dx * dx + dy * dy));
            b *= 0.5f;
            px = .8f * px * 2.0f; py = -.6f * py * 2.0f;
        }
        V4 o;
        o.x = o.y = o.z = 1.0f;
        o.w = p.y - 250.0f * 120.0f * a;
        return o;
    }
    template <bool TRAPS> RM_DI V4 eval(V3 p) const {
        if (U.body == RM_BODY_COMPOSITE) {
            // cginc:349 on the world-space p (no wrapper): the exponent of the bulb differs from point to point
            const float e = menger<false>(rm_rotate_euler(Es, p)).w + 1.1f;
            return bulb<TRAPS, false>(rm_rotate_euler(Et, p), e);
        }
        const V3 l = local(p);
        V4 o;
        switch (U.body) {                                        // wave-uniform
        case RM_BODY_MENGER:     o = menger<TRAPS>(l); break;
        case RM_BODY_SIERPINSKI: o = sierpinski<TRAPS>(l); break;
        case RM_BODY_MANDELBULB: o = bulb<TRAPS, false>(l, U.f_power); break;
        case RM_BODY_TERRAIN:    o = terrain(l); break;
        default:                 o = bulb<TRAPS, true>(l, U.f_power); break;
        }
        if (!unit) o.w = o.w * P0.ms;
        return o;
    }
    RM_DI float w(V3 p) const { return eval<false>(p).w; }
    RM_DI float wm(V3 p) const { return w(p); }
    RM_DI V4 full(V3 p) const { return eval<true>(p); }
};

// Domain operators in front of any scene: the functions the reference defines for this and never calls.
//   repeat      p = shmod(p, m) on the axes with m > 0; mod = fmod made non-negative (cginc:84-99).  fmodf is
//               the exact remainder on both the host and the GPU, so CPU and GPU results agree bit for bit
//   boxfold     b *= .5; p = clamp(p, -b, b) * 2.0 - p                 (cginc:156-159)
//   spherefold  r = length(p); if (r < R) p = p * R * R / (r * r)      (cginc:161-165)
// The operator is wave-uniform.  Compiled into the kernels with the optional lines only.
template <class S>
struct SceneDom {
    const S &in;
    int op;
    float mx, my, mz, size;
    RM_DI SceneDom(const S &s, const RmParams &u) : in(s), op(u.dom_op), mx(u.dom_period[0]), my(u.dom_period[1]),
                                                   mz(u.dom_period[2]), size(u.dom_size) {}
    RM_DI static float shmod(float x, float m) {
        const float hm = m * .5f;
        float r = fmodf(x + hm, m);
        r = r < 0.0f ? r + m : r;
        return r - hm;
    }
    RM_DI static float fold1(float x, float b) { return fminf(fmaxf(x, -b), b) * 2.0f - x; }
    RM_DI V3 apply(V3 p) const {
        if (op == RM_DOM_REPEAT) {
            if (mx > 0.0f) p.x = shmod(p.x, mx);
            if (my > 0.0f) p.y = shmod(p.y, my);
            if (mz > 0.0f) p.z = shmod(p.z, mz);
        } else if (op == RM_DOM_BOXFOLD) {
            const float b = size * .5f;
            p = mk3(fold1(p.x, b), fold1(p.y, b), fold1(p.z, b));
        } else if (op == RM_DOM_SPHEREFOLD) {
            const float r = sqrtf(rm_dot3(p, p));
            if (r < size) {
                const float rr = r * r;
                p = mk3(p.x * size * size / rr, p.y * size * size / rr, p.z * size * size / rr);
            }
        }
        return p;
    }
    RM_DI float w(V3 p) const { return in.w(apply(p)); }
    RM_DI float wm(V3 p) const { return in.wm(apply(p)); }
    RM_DI V4 full(V3 p) const { return in.full(apply(p)); }
};

RM_DI uint32_t rm_unorm8(float x) {
    x = fminf(fmaxf(x, 0.0f), 1.0f);           // saturate; a NaN becomes 0
    return (uint32_t)floorf(x * 255.0f + 0.5f);
}
