#pragma once
// rm_fast.h -- the march loops' distance evaluation for the opt-in tolerance kernel (rm_set_kernel(ctx, 4)).
//
// Same formulas as rm_math.h (Assets/Raymarching.cginc:62-66, 110-113, 169-192, 334-364), but compiled for speed
// instead of for bit-equality with a strict binary32 evaluation:
//   * a*b + c is one fused multiply-add, written out with fma_() where it matters (left to the compiler, the SLP
//     vectoriser pairs the multiplies into v_pk_mul_f32 first and most of the fusing is lost: 233 us instead of 206 us
//     for the 3840x2160 frame; the strict kernel takes 255 us),
//   * length() takes the raw v_sqrt_f32 (within 1 ulp) instead of the correctly rounded 6-op root,
//   * a divide by a primitive's scale is a multiply by the host's reciprocal, without the residual correction.
// Only the two march loops (cginc:372-384, shader:108-120) use it.  The six normal taps and the colour evaluation
// stay on the strict path: the taps are a central difference with epsilon 5e-5, whose value is dominated by the
// rounding of the scene function itself.  Frames therefore differ from the strict kernels' in the last bits of
// the march length; the tests hold this kernel to the tolerance of the port (hit mask equal but for 0.01 % of the
// pixels, depth within 1e-4 relative, colour within 1/255 on 99.9 %), not to bit-equality.
#include "rm_math.h"

#pragma clang fp contract(fast)
namespace rmf {

RM_DI float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RM_DI float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RM_DI float len3(V3 a) { return fsqrt(fma_(a.z, a.z, fma_(a.y, a.y, a.x * a.x))); }
RM_DI float len2(float x, float y) { return fsqrt(fma_(y, y, x * x)); }

template <bool ROT, bool SCALED, class PT>
RM_DI V3 to_local(const PT &P, V3 p) {
    V3 v = mk3(p.x - P.px, p.y - P.py, p.z - P.pz);
    if (ROT) {
        const V3 u = mk3(P.ux, P.uy, P.uz);
        const float a = 2.0f * fma_(u.z, v.z, fma_(u.y, v.y, u.x * v.x));
        const V3 cr = mk3(fma_(u.y, v.z, -(u.z * v.y)), fma_(u.z, v.x, -(u.x * v.z)), fma_(u.x, v.y, -(u.y * v.x)));
        v = mk3(fma_(cr.x, P.w2, fma_(u.x, a, v.x * P.c)), fma_(cr.y, P.w2, fma_(u.y, a, v.y * P.c)), fma_(cr.z, P.w2, fma_(u.z, a, v.z * P.c)));
    }
    if (SCALED) v = mk3(v.x * P.rx, v.y * P.ry, v.z * P.rz);
    return v;
}

template <class PT> RM_DI float sphere(const PT &P, V3 p) { return len3(p) - P.a0; }
template <class PT> RM_DI float box(const PT &P, V3 p) {
    const float dx = fabsf(p.x) - P.a0, dy = fabsf(p.y) - P.a1, dz = fabsf(p.z) - P.a2;
    return len3(mk3(fmaxf(dx, 0.0f), fmaxf(dy, 0.0f), fmaxf(dz, 0.0f))) + fminf(fmaxf(dx, fmaxf(dy, dz)), 0.0f);
}
template <class PT> RM_DI float link(const PT &P, V3 p) {
    const float qy = fmaxf(fabsf(p.y) - P.a0, 0.0f);
    return len2(len2(p.x, qy) - P.a1, p.z) - P.a2;
}
template <class PT> RM_DI float cylinder(const PT &P, V3 p) {
    const float dx = len2(p.x, p.z) - P.a0, dy = fabsf(p.y) - P.a1;
    return fminf(fmaxf(dx, dy), 0.0f) + len2(fmaxf(dx, 0.0f), fmaxf(dy, 0.0f));
}
template <class PT> RM_DI float torus(const PT &P, V3 p) { return len2(len2(p.x, p.y) - P.a0, p.z) - P.a1; }

template <bool ROT, bool SCALED, class PT>
RM_DI float prim(const PT &P, int type, V3 p) {
    const V3 l = to_local<ROT, SCALED>(P, p);
    float d;
    switch (type) {
    case 0:  d = sphere(P, l); break;
    case 1:  d = box(P, l); break;
    case 2:  d = link(P, l); break;
    case 3:  d = cylinder(P, l); break;
    default: d = torus(P, l); break;
    }
    return SCALED ? d * P.ms : d;
}

RM_DI float smin_w(float a, float b) {
    const float e = b - a;
    const float h = rm_clamp01(0.5f + e);
    return fma_(-(0.5f * h), 1.0f - h, fma_(-h, e, b));
}

// The reference's four slots with the march-loop evaluation on the fast path; everything else (w for the normal
// taps, full for the colour) is the strict Scene4's.  SCL is RM_SCL_UNIT or RM_SCL_RCP (the host maps scenes that
// need a true division to RM_SCL_RCP here: the reciprocals are always at hand).
template <bool ROT, int SCL, int SHAPE = 0>
struct Scene4Fast : Scene4<ROT, SCL> {
    RM_DI explicit Scene4Fast(const RmParams &u) : Scene4<ROT, SCL>(u) {}
    RM_DI float wm(V3 p) const {
        const float o0 = prim<ROT, SCL != RM_SCL_UNIT>(this->P0, 0, p);
        const float o1 = prim<ROT, SCL != RM_SCL_UNIT>(this->P1, 1, p);
        const float o2 = prim<ROT, SCL != RM_SCL_UNIT>(this->P2, 2, p);
        const float o3 = prim<ROT, SCL != RM_SCL_UNIT>(this->P3, 3, p);
        return smin_w(smin_w(o0, o1), smin_w(o2, o3));
    }
};

}  // namespace rmf
#pragma clang fp contract(off)

template <bool ROT, int SCL, int SHAPE> struct rm_is_fast<rmf::Scene4Fast<ROT, SCL, SHAPE>> { static constexpr bool value = true; };
template <bool ROT, int SCL, int SHAPE> struct rm_shape<rmf::Scene4Fast<ROT, SCL, SHAPE>> { static constexpr int value = SHAPE; };
