#pragma once
// rm_kernels_common.h -- gfx950 kernels of librmhip: the fused per-pixel march
// (primary ray + central-difference normal + colour + soft-shadow march + shade)
// that replaces pass 0 of the reference's fragment shader
// (Assets/Raymarching.shader:149-216 with Assets/Raymarching.cginc:334-392), and the
// depth-only passes built from raymarch().
#include <hip/hip_runtime.h>
#include "rm_device.h"
#include "rm_math.h"
#include "rm_launch.h"

// block geometry of the tile kernel: RM_BWX x RM_BWY waves, each an 8x8 pixel tile
#ifndef RM_BWX
#define RM_BWX 2
#endif
#ifndef RM_BWY
#define RM_BWY 2
#endif
#define RM_BLOCK (64 * RM_BWX * RM_BWY)
#ifndef RM_HEAVY_PRIO
#define RM_HEAVY_PRIO 3
#endif
// a tile whose cost is this many times the heavy threshold also puts its eight neighbours on the next frame's heavy list
// (0: never)
#ifndef RM_HEAVY_DILATE
#define RM_HEAVY_DILATE 2
#endif
// (RM_SLOT_WORDS, the words per block in the counter-slot buffer, is defined in rm_device.h; in a timeline build words 8..15
// are the wave's last four tiles: start time in 10 ns units after the wave's start, and cost << 20 | queue entry)
#ifdef RM_MIN_WAVES
#define RM_LB __launch_bounds__(RM_BLOCK, RM_MIN_WAVES)
#else
#define RM_LB __launch_bounds__(RM_BLOCK)
#endif

namespace {

struct MarchResult { bool hit; int evals; float length; float contact; int steps; };

// cginc:367-392.  `n_steps` = number of i >= 0 with (float)i < min(2000,_MaxSteps), `depth` already
// min'ed with _MaxDist.  The loop only carries dm; whether it ended on the contact test is
// re-derived from the last distance after the loop (same comparison, same operands).
template <class S>
RM_DI MarchResult rm_raymarch(const S &sc, V3 ro, V3 rd, float depth, int n_steps, float sf) {
    const float kc = rm_vgpr(.0025f);
    MarchResult r;
    r.evals = 0;
    float dm = 0.0f, adts = __builtin_inff(), contact = 0.0f;
    constexpr bool FAST = rm_is_fast<S>::value;
    for (int i = 0; i < n_steps; i++) {
        V3 cp = FAST ? mk3(__builtin_fmaf(rd.x, dm, ro.x), __builtin_fmaf(rd.y, dm, ro.y), __builtin_fmaf(rd.z, dm, ro.z))
                     : mk3(ro.x + rd.x * dm, ro.y + rd.y * dm, ro.z + rd.z * dm);
        adts = fabsf(sc.wm(cp));
        r.evals++;
        dm = FAST ? __builtin_fmaf(adts, sf, dm) : dm + adts * sf;
        contact = dm * kc;
        if (adts < contact) break;
        if (dm > depth) break;
    }
    r.hit = adts < contact;
    r.contact = contact;
    r.length = dm;
    // r.steps = the loop counter after the loop (cginc:387): one less than the evaluations if the loop
    // was left by a break
    r.steps = (r.evals > 0 && (adts < contact || dm > depth)) ? r.evals - 1 : r.evals;
    return r;
}

// shader:102-123 with k = 16
//
// Exact skip of the left pair (scene types with rm_pair_skip).  The root of the 4-slot scene is smin(smin(o0,o1), smin(o2,o3))
// (cginc:363), evaluated as e = b - a; h = clamp(.5 + e); (b - h*e) - (.5*h)*(1 - h): once the left operand a lies .5 above the
// right operand b, h is 0 and the value is (b - (-0)) - 0 = b + 0, whatever a is.  Every primitive is a Euclidean distance
// function times min(scale) of a point divided by scale, so with positive scales it -- and any smin of such -- changes by at most
// the distance the sample point moves.  The march therefore keeps a lower bound of the left pair: its last evaluated value less
// |L| times the march distance covered since, less 2^-21 of the largest coordinate the march can reach per step (the sample
// points are fp32 sums: a point may move an ulp further than the march distance says).  While that bound stays .51 (+ 1e-4 of
// its magnitude: the rounding of the bound and of the evaluations) above the right pair's value in EVERY lane of the wave
// still marching, the left pair (sphere, box: 46 of the ~100 instructions of an evaluation) is not evaluated.  Marches that
// can reach coordinates beyond 1000 never skip (the .01 of slack is sized for evaluation errors below that).  Shadow rays
// that slide along the link or the cylinder -- the longest dependent chains of a frame of the sample scene -- run on the
// right pair alone; on samples of the 1080p frame's tiles between a fifth and three fifths of the shadow wave steps skip
// (primary marches: 12-15 %, which does not pay for the vote there), with every distance, step count and depth bit unchanged
// (tests/dev/dev_pair_skip_sim.py).
template <class S>
RM_DI float rm_getshadow(const S &sc, V3 p, V3 n, V3 L, float contact, float depth, int n_steps, float sf, int &evals) {
    const float kc = rm_vgpr(.0025f);
    float dm = 0.0f, res = 1.0f;
    p = mk3(p.x + (n.x * contact) * 2.0f, p.y + (n.y * contact) * 2.0f, p.z + (n.z * contact) * 2.0f);
    int ev = 0;
    constexpr bool FAST = rm_is_fast<S>::value;
    if constexpr (rm_pair_skip<S>::value && !FAST) {
        const bool may_skip = sc.lipschitz();
        const float len0 = sqrtf(rm_dot3(L, L)) * 1.0011f;       // the sample point moves by at most len0 * (step in dm); the host admits quaternions up to |q|^2 = 1.001
        const float cmax = fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z)) + len0 * depth;     // no coordinate of a sample point exceeds this
        const float len = cmax < 1000.0f ? len0 : __builtin_inff();                             // (NaN included: the bound then never holds)
        const float eps = cmax * 4.76837158203125e-07f;          // 2^-21 of it: four ulps
        float lb = -__builtin_inff();                            // lower bound of left(cp); -inf: unknown
        for (int i = 0; i < n_steps; i++) {
            const V3 cp = mk3(p.x + L.x * dm, p.y + L.y * dm, p.z + L.z * dm);
            const float b = sc.right(cp);
            float dts;
            if (__all(lb - b >= __builtin_fmaf(fabsf(lb), 1e-4f, 0.51f))) {
                dts = b + 0.0f;                                  // rm_smin_w(a, b) for any a >= b + .5, bit for bit
            } else {
                const float a = sc.left(cp);
                dts = rm_smin_w(a, b);
                if (may_skip) lb = a;
            }
            ev++;
            res = fminf(res, 16.0f * dts / dm);
            const float inc = fabsf(dts) * sf;
            dm = dm + inc;
            lb = (lb - inc * len) - eps;
            contact = dm * kc;
            if (fabsf(dts) < contact) break;
            if (dm > depth) break;
        }
        evals = ev;
        return res;
    }
    for (int i = 0; i < n_steps; i++) {
        V3 cp = FAST ? mk3(__builtin_fmaf(L.x, dm, p.x), __builtin_fmaf(L.y, dm, p.y), __builtin_fmaf(L.z, dm, p.z))
                     : mk3(p.x + L.x * dm, p.y + L.y * dm, p.z + L.z * dm);
        float dts = sc.wm(cp);
        ev++;
        // (fast: v_rcp_f32 is within 1 ulp; at dm == 0 it gives inf, and dts * inf is the +-inf / NaN that dts / 0 is)
        res = fminf(res, FAST ? (16.0f * dts) * __builtin_amdgcn_rcpf(dm) : 16.0f * dts / dm);
        dm = FAST ? __builtin_fmaf(fabsf(dts), sf, dm) : dm + fabsf(dts) * sf;
        contact = dm * kc;
        if (fabsf(dts) < contact) break;
        if (dm > depth) break;
    }
    evals = ev;
    return res;
}

// pointer into LDS with its address space spelled out: through a generic pointer the reads would be flat loads
#define RM_LDS __attribute__((address_space(3)))
typedef const volatile RM_LDS float *rm_lds_cfloat;

struct PixelOut { uint32_t rgba; float depth; uint32_t hit; size_t oidx; uint32_t steps; };

// Ray of pixel (x, local row yl, pose): origin, un-normalised direction, march depth limit,
// and the pixel's index in the full frame (inputs) and in the compact output.
// `sv`: optional copy of U.view in LDS (ro, bl, br, tl, tr as in RmView).  The persistent kernel reads the
// view from there once per tile: held in scalar registers across the whole tile loop, these 15 values
// (with everything else that loop keeps) overflow the SGPR file and come back as v_readlane traffic.
template <int SHAPE = 0>
RM_DI void rm_setup_ray(const RmParams &U, int x, int yl, int pose, V3 &ro, V3 &rd, float &depth, size_t &gidx, size_t &oidx,
                        rm_lds_cfloat sv = nullptr, bool use_sv = false) {
    constexpr bool PLAIN = SHAPE != 0;
    // compact local row -> frame row (row-band shard; identity when n_parts == 1)
    int y = yl;
    if (!PLAIN && U.n_parts != 1) {
        const int band = yl / U.band_rows;
        y = (band * U.n_parts + U.part) * U.band_rows + (yl - band * U.band_rows);
    }
    gidx = (size_t)y * (size_t)U.w + (size_t)x;
    oidx = ((size_t)pose * (size_t)U.rows + (size_t)yl) * (size_t)U.w + (size_t)x;

    const RmView &V = (SHAPE == 2 || (!PLAIN && U.views)) ? U.views[pose] : U.view;
    // pixel centre / frame size; the host has checked for this frame size that the 3-op exact division
    // by a uniform divisor gives the IEEE quotient for every column and row, else inv_w = 0
    float uu, vv;
    if (PLAIN || U.inv_w != 0.0f) {
        uu = rm_div_by((float)x + 0.5f, (float)U.w, U.inv_w);
        vv = rm_div_by((float)y + 0.5f, (float)U.h, U.inv_h);
    } else {
        uu = ((float)x + 0.5f) / (float)U.w;
        vv = ((float)y + 0.5f) / (float)U.h;
    }
    if (!PLAIN && U.mode == RM_MODE_DEPTH_ORTHO) {
        V3 c0 = mk3(U.ortho_c[0][0], U.ortho_c[0][1], U.ortho_c[0][2]);
        V3 c1 = mk3(U.ortho_c[1][0], U.ortho_c[1][1], U.ortho_c[1][2]);
        V3 c2 = mk3(U.ortho_c[2][0], U.ortho_c[2][1], U.ortho_c[2][2]);
        V3 c3 = mk3(U.ortho_c[3][0], U.ortho_c[3][1], U.ortho_c[3][2]);
        ro = rm_lerp3(rm_lerp3(c3, c2, uu), rm_lerp3(c0, c1, uu), vv);
        rd = mk3(U.ortho_fwd[0], U.ortho_fwd[1], U.ortho_fwd[2]);
    } else {
        // rasteriser interpolation of vert's viewDir (cginc:26-33) to the pixel centre
        if (SHAPE == 1 || SHAPE == 3 || SHAPE == 4 || (!PLAIN && use_sv)) {
            typedef float rm_f4 __attribute__((ext_vector_type(4)));
            const volatile RM_LDS rm_f4 *q = (const volatile RM_LDS rm_f4 *)sv;      // four ds_read_b128, one wait
            const rm_f4 a = q[0], b = q[1], c = q[2], d = q[3];
            ro = mk3(a.x, a.y, a.z);
            const V3 bl = mk3(a.w, b.x, b.y), br = mk3(b.z, b.w, c.x);
            const V3 tl = mk3(c.y, c.z, c.w), tr = mk3(d.x, d.y, d.z);
            rd = rm_lerp3(rm_lerp3(bl, br, uu), rm_lerp3(tl, tr, uu), vv);
        } else {
            V3 bl = mk3(V.bl[0], V.bl[1], V.bl[2]), br = mk3(V.br[0], V.br[1], V.br[2]);
            V3 tl = mk3(V.tl[0], V.tl[1], V.tl[2]), tr = mk3(V.tr[0], V.tr[1], V.tr[2]);
            ro = mk3(rm_vgpr(V.ro[0]), rm_vgpr(V.ro[1]), rm_vgpr(V.ro[2]));
            rd = rm_lerp3(rm_lerp3(bl, br, uu), rm_lerp3(tl, tr, uu), vv);
        }
    }
    depth = U.far_plane;                                          // shader:154
    if (SHAPE == 3 || (!PLAIN && U.mode == RM_MODE_FRAG && U.sdepth)) depth = U.sdepth[gidx];
    depth = fminf(depth, U.max_dist);                             // cginc:368
}


// Shadow-map lookup that takes getshadow()'s place when U.shadow_mode != 0: 1 = lit, 0 = in shadow.  q is the point
// getshadow() would start from (shader:107).  Perspective map: c = mul(_ShadowMapVPMatrix, float4(q,1)), uv = c.xy/c.w*.5+.5,
// eye depth c.w against the stored r.length; directional map: coordinates along the rectangle's edges and its forward.
RM_DI float rm_shadow_lookup(const RmParams &U, V3 p, V3 n, float contact) {
    const V3 q = mk3(p.x + (n.x * contact) * 2.0f, p.y + (n.y * contact) * 2.0f, p.z + (n.z * contact) * 2.0f);
    float uu, vv, d;
    if (U.shadow_mode == 1) {
        const float cx = ((U.sm_row0[0] * q.x + U.sm_row0[1] * q.y) + U.sm_row0[2] * q.z) + U.sm_row0[3];
        const float cy = ((U.sm_row1[0] * q.x + U.sm_row1[1] * q.y) + U.sm_row1[2] * q.z) + U.sm_row1[3];
        const float cw = ((U.sm_row3[0] * q.x + U.sm_row3[1] * q.y) + U.sm_row3[2] * q.z) + U.sm_row3[3];
        if (!(cw > 0.0f)) return 1.0f;
        uu = (cx / cw) * .5f + .5f;
        vv = (cy / cw) * .5f + .5f;
        d = cw;
    } else {
        const V3 r = mk3(q.x - U.sm_bl[0], q.y - U.sm_bl[1], q.z - U.sm_bl[2]);
        uu = rm_dot3(r, mk3(U.sm_e1[0], U.sm_e1[1], U.sm_e1[2])) / U.sm_n1;
        vv = rm_dot3(r, mk3(U.sm_e2[0], U.sm_e2[1], U.sm_e2[2])) / U.sm_n2;
        d = rm_dot3(r, mk3(U.sm_fwd[0], U.sm_fwd[1], U.sm_fwd[2]));
    }
    if (!(uu >= 0.0f && uu < 1.0f && vv >= 0.0f && vv < 1.0f)) return 1.0f;
    int ix = (int)floorf(uu * (float)U.sm_w), iy = (int)floorf(vv * (float)U.sm_h);
    ix = ix > U.sm_w - 1 ? U.sm_w - 1 : ix;
    iy = iy > U.sm_h - 1 ? U.sm_h - 1 : iy;
    const float stored = U.shadow_map[(size_t)iy * (size_t)U.sm_w + (size_t)ix];
    return d > stored * U.sm_onebias ? 0.0f : 1.0f;
}


// The whole fragment program for one pixel (x, local row yl, pose): primary march, and for a
// hit pixel the six normal taps, the colour evaluation, the shadow march and the shade
// (shader:149-216).  Miss pixels skip normal/shadow: the shader computes and then discards
// them (shader:197-202).  Adds this pixel's counts to c_hit / c_p / c_s.  Nothing is
// stored here: the caller decides when the framebuffer writes are issued.
template <class S>
RM_DI PixelOut rm_compute_pixel(const RmParams &U, const S &sc, int x, int yl, int pose,
                                unsigned &c_hit, unsigned &c_p, unsigned &c_s, rm_lds_cfloat sv = nullptr, bool use_sv = false) {
    PixelOut out;
    V3 ro, rd;
    float depth;
    size_t gidx;
    constexpr bool PLAIN = rm_shape<S>::value != 0;
    rm_setup_ray<rm_shape<S>::value>(U, x, yl, pose, ro, rd, depth, gidx, out.oidx, sv, use_sv);

    const float sf = rm_vgpr(U.step_factor);
    MarchResult r = rm_raymarch(sc, ro, rd, depth, U.n_steps, sf);
    c_p += (unsigned)r.evals;
    c_hit += r.hit ? 1u : 0u;
    out.depth = r.length;
    out.hit = r.hit ? 1u : 0u;
    out.rgba = 0;
    out.steps = 0;

    if (PLAIN || U.mode == RM_MODE_FRAG) {
        if (r.hit) {
            const V3 L = mk3(rm_vgpr(U.light[0]), rm_vgpr(U.light[1]), rm_vgpr(U.light[2]));
            V3 hp = mk3(ro.x + rd.x * r.length, ro.y + rd.y * r.length, ro.z + rd.z * r.length);   // shader:174
            const float e = U.nss;
            // shader:63-67: central differences along x, y, z.  Kept as a rolled loop: unrolled, the
            // scheduler interleaves all six evaluations and the register file overflows into scratch.
            V3 raw = mk3(0.0f, 0.0f, 0.0f);
#pragma nounroll
            for (int a = 0; a < 3; a++) {
                const V3 o = mk3(a == 0 ? e : 0.0f, a == 1 ? e : 0.0f, a == 2 ? e : 0.0f);
                const float d = sc.w(mk3(hp.x + o.x, hp.y + o.y, hp.z + o.z)) - sc.w(mk3(hp.x - o.x, hp.y - o.y, hp.z - o.z));
                raw.x = a == 0 ? d : raw.x; raw.y = a == 1 ? d : raw.y; raw.z = a == 2 ? d : raw.z;
            }
            V3 n = rm_normalize(raw);
            V4 s = sc.full(hp);                                                                   // shader:193
            int sev = 0;
            float sh;
            if (rm_shape<S>::value == 4) sh = rm_shadow_lookup(U, hp, n, r.contact);       // the bound _ShadowMap answers instead of the march
            else sh = rm_getshadow(sc, hp, n, L, r.contact, U.shadow_depth, U.n_steps, sf, sev);
            c_s += (unsigned)sev;
            float shadow = sh * 0.75f + .25f;                        // remap(x,0,1,.25,1), cginc:80-82
            float ndl = rm_dot3(n, L) * .5f + .5f;
            float ao = rm_clamp01(rm_len3(mk3(raw.x / U.nss_raw, raw.y / U.nss_raw, raw.z / U.nss_raw)));  // shader:125-127
            float cr = ((s.x * ndl) * shadow) * ao;
            float cg = ((s.y * ndl) * shadow) * ao;
            float cb = ((s.z * ndl) * shadow) * ao;
            out.rgba = rm_unorm8(cr) | (rm_unorm8(cg) << 8) | (rm_unorm8(cb) << 16) | 0xFF000000u;
        } else if (rm_shape<S>::value == 3 || (!PLAIN && U.src)) {
            out.rgba = reinterpret_cast<const uint32_t *>(U.src)[gidx] | 0xFF000000u;
        } else {
            out.rgba = U.bg_rgba8;
        }
    }
    return out;
}

// frag with the lines the reference keeps commented out switched on by U.fx_flags / U.fx_view:
// the one-bounce loop (shader:178-187), the alternative outputs (:198-201), the step-count term
// (:203) and the light integration along the ray (:206-212).  Used only by the *_fx kernels, so the
// default kernels carry none of this.  depth / mask / counters describe the camera ray.
RM_DI V3 rm_reflect(V3 i, V3 n) { const float t = 2.0f * rm_dot3(i, n); return mk3(i.x - n.x * t, i.y - n.y * t, i.z - n.z * t); }
RM_DI V3 rm_refract(V3 i, V3 n, float eta) {
    const float d = rm_dot3(n, i);
    const float k = 1.0f - (eta * eta) * (1.0f - d * d);
    if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    const float t = eta * d + sqrtf(k);
    return mk3(i.x * eta - n.x * t, i.y * eta - n.y * t, i.z * eta - n.z * t);
}

// shader:142-147
RM_DI V3 rm_refract2(V3 i, V3 n, float eta) {
    eta = 2.0f - eta;
    const float cosi = rm_dot3(n, i);
    const float t = -cosi + eta * cosi;
    return mk3(i.x * eta - n.x * t, i.y * eta - n.y * t, i.z * eta - n.z * t);
}
// shader:84-100 with the light of shader:157-162 (range 1000, intensity 1)
RM_DI float rm_getbrightness(V3 p, V3 n, V3 lp, bool hard) {
    const float range = 1000.0f, intensity = 1.0f;
    const float dist = sqrtf(rm_dot3(mk3(p.x - lp.x, p.y - lp.y, p.z - lp.z), mk3(p.x - lp.x, p.y - lp.y, p.z - lp.z)));
    if (dist >= range) return 0.0f;
    const float dn = rm_dot3(n, mk3(lp.x - p.x, lp.y - p.y, lp.z - p.z));
    if (hard) return rm_lerp(rm_clamp01(dn) * intensity, 0.0f, dist / range);
    float d = fminf(fmaxf(dn, -1.0f), 1.0f) * .5f + .5f;
    d *= d;
    return rm_lerp(d * intensity, 0.0f, dist / range);
}
// shader:129-140; atan(x) = the canonical atan2(x, 1)
RM_DI float rm_getscatter(V3 ro, V3 rd, float length, V3 lp) {
    const V3 q = mk3(ro.x - lp.x, ro.y - lp.y, ro.z - lp.z);
    const float b = rm_dot3(rd, q), c = rm_dot3(q, q);
    const float s = 1.0f / sqrtf(c - b * b);
    return s * (rm_c_atan2((length + b) * s, 1.0f) - rm_c_atan2(b * s, 1.0f));
}

template <class S>
RM_DI V3 rm_normal_raw(const RmParams &U, const S &sc, V3 hp) {
    const float e = U.nss;
    V3 raw = mk3(0.0f, 0.0f, 0.0f);
#pragma nounroll
    for (int a = 0; a < 3; a++) {
        const V3 o = mk3(a == 0 ? e : 0.0f, a == 1 ? e : 0.0f, a == 2 ? e : 0.0f);
        const float d = sc.w(mk3(hp.x + o.x, hp.y + o.y, hp.z + o.z)) - sc.w(mk3(hp.x - o.x, hp.y - o.y, hp.z - o.z));
        raw.x = a == 0 ? d : raw.x; raw.y = a == 1 ? d : raw.y; raw.z = a == 2 ? d : raw.z;
    }
    return raw;
}

template <class S>
RM_DI PixelOut rm_compute_pixel_fx(const RmParams &U, const S &sc, int x, int yl, int pose,
                                   unsigned &c_hit, unsigned &c_p, unsigned &c_s) {
    PixelOut out;
    V3 ro, rd;
    float depth;
    size_t gidx;
    rm_setup_ray(U, x, yl, pose, ro, rd, depth, gidx, out.oidx);
    const float sf = rm_vgpr(U.step_factor);
    MarchResult r = rm_raymarch(sc, ro, rd, depth, U.n_steps, sf);
    c_p += (unsigned)r.evals;
    c_hit += r.hit ? 1u : 0u;
    out.depth = r.length;
    out.hit = r.hit ? 1u : 0u;
    out.rgba = 0;
    out.steps = 0;
    if (U.mode != RM_MODE_FRAG) return out;

    V3 col;
    if (U.src) {
        const uint32_t t = reinterpret_cast<const uint32_t *>(U.src)[gidx];
        col = mk3((float)(t & 255u) / 255.0f, (float)((t >> 8) & 255u) / 255.0f, (float)((t >> 16) & 255u) / 255.0f);
    } else {
        col = mk3(U.bg[0], U.bg[1], U.bg[2]);
    }
    const V3 L = mk3(rm_vgpr(U.light[0]), rm_vgpr(U.light[1]), rm_vgpr(U.light[2]));
    if (r.hit) {
        V3 hp = mk3(ro.x + rd.x * r.length, ro.y + rd.y * r.length, ro.z + rd.z * r.length);
        V3 raw = rm_normal_raw(U, sc, hp);
        V3 n = rm_normalize(raw);
        if (U.fx_flags & RM_FX_BOUNCE) {
            ro = mk3(hp.x - (n.x * r.contact) * 2.0f, hp.y - (n.y * r.contact) * 2.0f, hp.z - (n.z * r.contact) * 2.0f);
            rd = sc.w(hp) >= 0.0f ? ((U.fx_flags & RM_FX_REFRACT2) ? rm_refract2(rd, n, U.refr_index) : rm_refract(rd, n, 1.1f)) : rm_reflect(rd, n);
            r = rm_raymarch(sc, ro, rd, fminf(2000.0f, U.max_dist), U.n_steps, sf);
            hp = mk3(ro.x + rd.x * r.length, ro.y + rd.y * r.length, ro.z + rd.z * r.length);
            raw = rm_normal_raw(U, sc, hp);
            n = rm_normalize(raw);
        }
        V4 s = sc.full(hp);
        int sev = 0;
        float sh;
        if (U.shadow_mode != 0) sh = rm_shadow_lookup(U, hp, n, r.contact);
        else sh = rm_getshadow(sc, hp, n, L, r.contact, U.shadow_depth, U.n_steps, sf, sev);
        c_s += (unsigned)sev;
        if (r.hit) {                                           // the bounced ray may have missed: then tex stays
            float shadow = sh * 0.75f + .25f;
            float ndl = rm_dot3(n, L) * .5f + .5f;
            if (U.light_model != 0) ndl = rm_getbrightness(hp, n, mk3(U.light_pos[0], U.light_pos[1], U.light_pos[2]), U.light_model == 2);
            float ao = rm_clamp01(rm_len3(mk3(raw.x / U.nss_raw, raw.y / U.nss_raw, raw.z / U.nss_raw)));
            col = mk3(((s.x * ndl) * shadow) * ao, ((s.y * ndl) * shadow) * ao, ((s.z * ndl) * shadow) * ao);
            if (U.fx_view == 1) col = mk3(shadow, shadow, shadow);
            if (U.fx_view == 2) col = mk3(n.x * .5f + .5f, n.y * .5f + .5f, n.z * .5f + .5f);
            if (U.fx_view == 3) { const float v = rm_clamp01(s.x * 2.5f - 1.75f); col = mk3(v, v, v); }
        }
    }
    if (U.fx_flags & RM_FX_STEP_HEAT) {
        const float h = (float)r.steps / 100.0f * U.step_factor;
        col = mk3(col.x + h, col.y + h, col.z + h);
    }
    if (U.fx_flags & RM_FX_SCATTER) {
        const float sct = rm_getscatter(ro, rd, r.length, mk3(U.light_pos[0], U.light_pos[1], U.light_pos[2]));
        col = mk3(col.x + sct, col.y + sct, col.z + sct);
    }
    if (U.fx_flags & RM_FX_VOLUMETRIC) {
        float la = 0.0f;
        const float lim = fminf(5.0f, r.length);
        const V3 zero = mk3(0.0f, 0.0f, 0.0f);
        for (float i = 0.0f; i * .1f < lim; i += 1.0f) {
            const V3 rp = mk3(ro.x + (rd.x * i) * .1f, ro.y + (rd.y * i) * .1f, ro.z + (rd.z * i) * .1f);
            int ev = 0;
            la += rm_getshadow(sc, rp, zero, L, r.contact, U.shadow_depth, U.n_steps, sf, ev) * .05f;
        }
        col = mk3(col.x + la, col.y + la, col.z + la);
    }
    out.rgba = rm_unorm8(col.x) | (rm_unorm8(col.y) << 8) | (rm_unorm8(col.z) << 16) | 0xFF000000u;
    return out;
}

// one dword of colour, one of depth and one byte of mask per pixel; a wave writes an 8x8 tile
// as eight 32-byte row segments per plane
template <bool PLAIN = false>
RM_DI void rm_store_pixel(const RmParams &U, const PixelOut &o) {
    if (PLAIN || (U.mode == RM_MODE_FRAG && U.dst_rgba)) reinterpret_cast<uint32_t *>(U.dst_rgba)[o.oidx] = o.rgba;
    if (PLAIN || U.dst_depth) U.dst_depth[o.oidx] = o.depth;
    if (PLAIN || U.dst_hit) U.dst_hit[o.oidx] = (uint8_t)o.hit;
}
// the optional per-pixel step-count plane (the reference's nearest thing is the commented-out heat map, shader:203)
RM_DI void rm_store_steps(const RmParams &U, const PixelOut &o) {
    if (U.dst_steps) U.dst_steps[o.oidx] = o.steps;
}

// wave-wide maximum, returned as a uniform value: a butterfly inside each 16-lane row with DPP
// modifiers (no LDS crossbar, no waitcnt), then one lane per row read back and combined on the
// scalar unit.  Must be called with all 64 lanes active.
RM_DI unsigned rm_wave_max(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  v = v > t ? v : t;   // quad_perm [1,0,3,2]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  v = v > t ? v : t;   // quad_perm [2,3,0,1]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); v = v > t ? v : t;   // row_half_mirror
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); v = v > t ? v : t;   // row_mirror
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

RM_DI unsigned rm_wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- kernel A: one launch block per image tile -------------------------------------------
// One thread = one pixel.  A wave covers an 8x8 pixel tile (neighbouring rays take similar
// step counts), a block RM_BWX x RM_BWY such tiles.  blockIdx.z = pose.
template <bool FX = false, class S>
RM_DI void rm_pixel(const RmParams &U, const S &sc) {
    __shared__ unsigned int s_cnt[3];
    const int tid = threadIdx.x;
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();
    unsigned c_hit = 0, c_p = 0, c_s = 0;
#ifdef RM_TIMELINE
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = tid & 63, wv = tid >> 6;
    const int x  = blockIdx.x * (8 * RM_BWX) + (wv % RM_BWX) * 8 + (lane & 7);
    const int yl = blockIdx.y * (8 * RM_BWY) + (wv / RM_BWX) * 8 + (lane >> 3);
    if (x < U.w && yl < U.rows) {
        PixelOut po = FX ? rm_compute_pixel_fx(U, sc, x, yl, (int)blockIdx.z, c_hit, c_p, c_s)
                         : rm_compute_pixel(U, sc, x, yl, (int)blockIdx.z, c_hit, c_p, c_s);
        po.steps = c_p | (c_s << 16);          // this thread's only pixel: the counters are the pixel's own (<= 2000 each)
        rm_store_pixel(U, po);
        rm_store_steps(U, po);
    }

    // Counters: per-wave sums by cross-lane shuffles, per-block sums in LDS, then ONE plain
    // store per counter into this block's own slot.  No global atomics: same-address atomics
    // from thousands of short blocks serialise in L2 (~5 ns each) and would bound the frame.
    c_hit = rm_wave_sum(c_hit); c_p = rm_wave_sum(c_p); c_s = rm_wave_sum(c_s);
    if (lane == 0) {
        if (c_hit) atomicAdd(&s_cnt[0], c_hit);
        if (c_p) atomicAdd(&s_cnt[1], c_p);
        if (c_s) atomicAdd(&s_cnt[2], c_s);
    }
    __syncthreads();
    if (tid < 3) {
        const unsigned bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        U.partials[RM_SLOT_WORDS * bid + tid] = s_cnt[tid];
#ifdef RM_TIMELINE
        if (tid == 0) {
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned *o = U.partials + RM_SLOT_WORDS * bid;
            o[3] = (unsigned)t_start; o[4] = (unsigned)(t_start >> 32); o[5] = (unsigned)t_end; o[6] = (unsigned)(t_end >> 32);
            o[7] = (hwid & 0xffffu) | (xcc << 16);
        }
#endif
    }
}

// ---- kernel B: persistent waves pulling 8x8 tiles from an ordered queue --------------------
// The grid is sized to what the chip holds at once; every wave loops: take a queue entry,
// render that tile, take the next.  The queue lists the tiles that were expensive in the
// previous frame of this context first, then all tiles in a scattered image order, skipping
// the ones already listed.  A frame's critical path is its most expensive tile (a few hundred
// dependent scene evaluations); starting those tiles at t = 0 on waves of their own hides them
// behind the bulk of cheap tiles instead of leaving them as a tail.  Each tile is rendered
// exactly once whatever the order, so the image does not depend on the history.
//
// Queue layout: [heavy tiles of the previous frame][static part][dynamic part].
//  * wave w < min(n_heavy, n_waves) starts with heavy entry w, the others with their static share: every
//    wave begins at t = 0 without touching a counter;
//  * the static part (Q.static_per_wave entries per non-heavy wave, dealt cyclically) needs no
//    atomics.  A device-scope returning atomic is a memory-side operation here (1-3 us, about
//    90 per us and word): far too slow to claim every 3-us tile that way;
//  * the dynamic part is claimed RM_Q_CHUNK entries (single frames: two) at a time from RM_Q_SHARDS counters, each
//    on its own 256-B line, by waves that have run out of static work; it evens out what the
//    static deal left unbalanced.  The claiming atomic is issued BEFORE the current tile is
//    rendered and its result is read after, ahead of the tile's framebuffer stores, so its
//    latency hides behind the march and never queues behind the stores.
//
// "Was this tile on the heavy list?" is answered from a bitmap of the previous frame's list,
// copied into LDS at kernel start (4 KB at 1080p).  Three bitmap/list sets rotate: read,
// appended for the next frame, cleared for the one after.
template <class S>
RM_DI void rm_queue_body(const RmParams &U, const S &sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned s_dyn[];
    unsigned *s_bitmap = s_dyn;      // the bitmap sits at LDS offset 0 (a compile-time address: a runtime offset here cost 8 %); a staged primitive table follows it
    const RmQueue &Q = U.queue;
    const int lane = threadIdx.x & 63;
    const unsigned gw = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);   // global wave id
    const unsigned n_waves = (gridDim.x * blockDim.x) >> 6;
    unsigned c_hit = 0, c_p = 0, c_s = 0;
#ifdef RM_TIMELINE
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    unsigned n_done = 0, max_cost = 0, t_wait = 0, n_claims = 0;
    unsigned tl_t[4] = {0, 0, 0, 0}, tl_c[4] = {0, 0, 0, 0}, tl_t0 = 0;
#endif

    unsigned n_heavy = 0;
    if (Q.use_history) {
        n_heavy = Q.heavy_count[Q.hv_cur];
        n_heavy = n_heavy < Q.heavy_cap ? n_heavy : Q.heavy_cap;
    }
    n_heavy = __builtin_amdgcn_readfirstlane(n_heavy);
    const unsigned n_entries = n_heavy + Q.n_tiles;
    const unsigned *heavy_cur = Q.heavy_list + (size_t)Q.hv_cur * Q.heavy_cap;
    unsigned *heavy_nxt = Q.heavy_list + (size_t)Q.hv_nxt * Q.heavy_cap;
    unsigned *bitmap_nxt = Q.bitmaps + (size_t)Q.hv_nxt * Q.bitmap_words;
    unsigned *counters = Q.counters + Q.cnt_cur * RM_Q_SHARDS * RM_Q_STRIDE;

    // frame-to-frame housekeeping: clear what the NEXT frame will append to / claim from (a self-contained launch
    // takes no part in that rotation: the host clears its counters in the stream ahead of it)
    if (Q.cnt_clr != RM_Q_SOLO) {
        if (gw == 0 && lane < RM_Q_SHARDS) Q.counters[(Q.cnt_clr * RM_Q_SHARDS + lane) * RM_Q_STRIDE] = 0u;
        if (gw == 0 && lane == 0) Q.heavy_count[Q.hv_clr] = 0u;
        unsigned *bitmap_clr = Q.bitmaps + (size_t)Q.hv_clr * Q.bitmap_words;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < Q.bitmap_words; i += gridDim.x * blockDim.x) bitmap_clr[i] = 0u;
    }
    // the view -> LDS (see rm_setup_ray)
    __shared__ __attribute__((aligned(16))) float s_view[16];
    if (threadIdx.x < 15) {
        const float *v = U.view.ro;               // ro, bl, br, tl, tr are consecutive in RmView
        s_view[threadIdx.x] = v[threadIdx.x];
    }
    rm_lds_cfloat sv = (rm_lds_cfloat)s_view;
    constexpr int SHAPE = rm_shape<S>::value;
    constexpr bool PLAIN = SHAPE != 0;
    const bool use_sv = SHAPE == 1 || SHAPE == 3 || SHAPE == 4 || (!PLAIN && U.views == nullptr);
    // this frame's heavy bitmap -> LDS
    if (Q.use_history) {
        const unsigned *bitmap_cur = Q.bitmaps + (size_t)Q.hv_cur * Q.bitmap_words;
        for (unsigned i = threadIdx.x; i < Q.bitmap_words; i += blockDim.x) s_bitmap[i] = bitmap_cur[i];
    }
    __syncthreads();

    // the first n_first heavy entries go one per wave; whatever the heavy list holds beyond the
    // number of waves is served, like the image-order entries, through the static and dynamic parts
    const unsigned n_first = n_heavy < n_waves ? n_heavy : n_waves;
    const unsigned n_static_waves = n_waves - n_first;
    const unsigned static_end = n_first + Q.static_per_wave * n_static_waves;      // <= n_entries by construction
    unsigned shard = blockIdx.x % RM_Q_SHARDS, dry = 0;      // dry: bit s set = shard s is exhausted
    unsigned static_left = gw >= n_first ? Q.static_per_wave : 0u;
    unsigned static_next = n_first + (gw - n_first);
    unsigned dyn_left = 0, dyn_next = 0;

    // entry -> tile id, or 0xffffffff for an entry that must be skipped
    auto entry_tile = [&](unsigned e) -> unsigned {
        if (e < n_heavy) return __builtin_amdgcn_readfirstlane(heavy_cur[e]);
        // image-order position -> tile through a fixed bijection (multiplication by a number
        // coprime with n_tiles): the tiles a wave gets from its cyclic static share are
        // scattered over the image instead of lining up in one image column
        unsigned t = (unsigned)(((unsigned long long)(e - n_heavy) * Q.scatter_mul) % Q.n_tiles);
        if (Q.use_history && ((s_bitmap[t >> 5] >> (t & 31)) & 1u)) return 0xffffffffu;   // rendered through the heavy list
        return t;
    };
    // next entry without an atomic, or 0xffffffff if the wave has to claim
    auto local_next = [&]() -> unsigned {
        if (static_left) { unsigned e = static_next; static_next += n_static_waves; static_left--; return e; }
        if (dyn_left) { dyn_left--; return dyn_next++; }
        return 0xffffffffu;
    };
    // Entries per claim: RM_Q_CHUNK, but two in the single-frame shapes.  A finer deal ends a frame more evenly (4K 235 -> 231 us,
    // 1080p 82 -> 81 us); a pose batch has eight times the tiles and pays for twice the claims instead (8 poses at 4K in one
    // launch: 230 -> 242 us per pose with two, 232 with three), so it keeps four.
    constexpr unsigned CHUNK = (SHAPE == 1 || SHAPE == 3 || SHAPE == 4) ? (RM_Q_CHUNK < 2 ? RM_Q_CHUNK : 2) : RM_Q_CHUNK;
    auto chunk_first = [&](unsigned k, unsigned s) -> unsigned { return static_end + (k * RM_Q_SHARDS + s) * CHUNK; };

    // A wave whose claim finds its shard exhausted goes on to the other shards only in pose batches and in launches too small
    // to have waves on every shard.  In a single frame the shards -- interleaved chunk lists over scattered tiles -- run dry
    // within a few chunks of each other, and each is drained by its own 512 waves; the looks and failed claims of 4096
    // waves touring the other seven were the frame's last 2 % (4K 240 -> 236 us, 1080p 83.5 -> 81.5 us, and two frames in
    // flight 224 -> 212 us per frame: the tour's memory-side operations also stood in the way of the other frame's claims).
    const bool steal = SHAPE == 2 || (!PLAIN && U.n_poses != 1) || gridDim.x < 4u * RM_Q_SHARDS;
    bool done = false;
    unsigned cur = gw < n_first ? gw : local_next();
    if (cur == 0xffffffffu) {
        // no static share (tiny frame): first claim, synchronously
        unsigned k = 0;
        if (lane == 0) k = atomicAdd(&counters[shard * RM_Q_STRIDE], 1u);
        k = __builtin_amdgcn_readfirstlane(k);
        cur = chunk_first(k, shard);
        if (cur >= n_entries) { cur = 0xffffffffu; dry |= 1u << shard; if (!steal) dry = (1u << RM_Q_SHARDS) - 1u; }
        else { unsigned last = cur + CHUNK < n_entries ? cur + CHUNK : n_entries; dyn_next = cur + 1; dyn_left = last - cur - 1; }
    }
    while (!done) {
        // 1. the entry after `cur`: from the wave's own stock, or by an atomic issued now and read in step 3
        unsigned nxt = local_next();
        bool claiming = false;
        unsigned k_raw = 0;
        // A wave about to render its first-round heavy tile does not claim ahead: the chunk would sit
        // reserved until that long tile is done and then run as a tail behind the frame's critical
        // chain (measured: 4 tiles = 12 us at the end of the 1080p frame).  It claims when it is through.
        const bool defer_claim = cur != 0xffffffffu && cur < n_first;
        // End game (this wave has already met an exhausted shard): look before claiming.  Every wave
        // has to learn that every shard is empty; as atomics those are thousands of same-address
        // operations that L2 serialises (~5 ns each, a 10-20 us tail on the frame), as loads they are
        // served in parallel.  The wave has nothing else to do at this point, so the loads are waited for.
        // (One shard at a time on purpose: with lane s looking at shard s, all eight in one round trip, every look of every
        // wave lands on every counter's line, and operations on one line queue up in the L2 behind the claims in flight:
        // +36 us per frame, measured.)
        while (nxt == 0xffffffffu && dry != 0u && dry != (1u << RM_Q_SHARDS) - 1u) {
            while (dry & (1u << shard)) shard = (shard + 1) % RM_Q_SHARDS;
            unsigned seen = 0;
            if (lane == 0) seen = __hip_atomic_load(&counters[shard * RM_Q_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = __builtin_amdgcn_readfirstlane(seen);
            if (chunk_first(seen, shard) < n_entries) break;       // something may be left there: claim it
            dry |= 1u << shard;
        }
        if (nxt == 0xffffffffu && !defer_claim && dry != (1u << RM_Q_SHARDS) - 1u) {
            while (dry & (1u << shard)) shard = (shard + 1) % RM_Q_SHARDS;
            if (lane == 0) k_raw = atomicAdd(&counters[shard * RM_Q_STRIDE], 1u);
            claiming = true;
        }
        // 2. render `cur` (registers only)
        PixelOut po;
        bool have = false;
        unsigned tile = cur == 0xffffffffu ? 0xffffffffu : entry_tile(cur);
        unsigned t_hit = 0, t_p = 0, t_s = 0;
#ifdef RM_TIMELINE
        tl_t0 = (unsigned)(__builtin_amdgcn_s_memrealtime() - t_start);
#endif
        if (tile != 0xffffffffu) {
            // a tile from the heavy list is a long dependent chain that the frame cannot end before:
            // its wave gets issue priority over waves rendering cheap tiles on the same SIMD
            if (cur < n_heavy) __builtin_amdgcn_s_setprio(RM_HEAVY_PRIO); else __builtin_amdgcn_s_setprio(0);
            unsigned pose = 0, t2 = tile;
            if (SHAPE == 2 || (!PLAIN && U.n_poses != 1)) {
                const unsigned per_pose = Q.tiles_x * Q.tiles_y;
                pose = tile / per_pose; t2 = tile - pose * per_pose;
            }
            const unsigned ty = (unsigned)(((unsigned long long)t2 * Q.tx_magic) >> 37), tx = t2 - ty * Q.tiles_x;
            const int x = (int)(tx * RM_TW) + (lane % RM_TW), yl = (int)(ty * RM_TH) + (lane / RM_TW);
            if (x < U.w && yl < U.rows) { po = rm_compute_pixel(U, sc, x, yl, (int)pose, t_hit, t_p, t_s, sv, use_sv); have = true; }
            c_hit += t_hit; c_p += t_p; c_s += t_s;
        }
        // 3. the claim has had the whole tile to come back; read it before the stores go out.  The register is read on every
        // path, claim or not: with the read under `if (claiming)` the compiler cannot see that an issued claim is always
        // consumed, keeps the register's pending write alive across the loop's back edge and puts an s_waitcnt vmcnt(0) ahead
        // of the next iteration's first write to it -- and since stores share that counter, every tile then began by waiting
        // for the acknowledgement of the previous tile's framebuffer stores.
#ifdef RM_TIMELINE
        const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
        const unsigned k_seen = __builtin_amdgcn_readfirstlane(k_raw);
        if (claiming) {
            unsigned k = k_seen;
#ifdef RM_TIMELINE
            asm volatile("" :: "s"(k));
            t_wait += (unsigned)(__builtin_amdgcn_s_memrealtime() - tw0); n_claims++;
#endif
            unsigned first = chunk_first(k, shard);
            if (first < n_entries) {
                unsigned last = first + CHUNK < n_entries ? first + CHUNK : n_entries;
                nxt = first; dyn_next = first + 1; dyn_left = last - first - 1;
            } else {
                dry |= 1u << shard;                       // try the next shard on the next round
                if (!steal) dry = (1u << RM_Q_SHARDS) - 1u;
                nxt = 0xffffffffu;
                if (dry == (1u << RM_Q_SHARDS) - 1u) done = true;
            }
        } else if (nxt == 0xffffffffu && !(defer_claim && dry != (1u << RM_Q_SHARDS) - 1u)) {
            done = true;
        }
        // 4. framebuffer stores of `cur`, and its cost for the next frame's ordering
        if (tile != 0xffffffffu) {
            if (have) rm_store_pixel<PLAIN>(U, po);
            // this lane's scene evaluations; the wave maximum is the tile's length
            const unsigned cost = rm_wave_max(t_p + t_s + (t_hit ? 7u : 0u));
#ifdef RM_TIMELINE
            n_done++; max_cost = max_cost > cost ? max_cost : cost;
            tl_t[0] = tl_t[1]; tl_t[1] = tl_t[2]; tl_t[2] = tl_t[3]; tl_t[3] = tl_t0;
            tl_c[0] = tl_c[1]; tl_c[1] = tl_c[2]; tl_c[2] = tl_c[3]; tl_c[3] = (cost << 20) | (cur & 0xfffffu);
#endif
#if RM_HEAVY_DILATE
            // A camera that moves carries a long tile into a neighbouring one by the next frame: a tile well above the
            // threshold also lists its eight neighbours (lane k < 9 takes neighbour k; the bitmap's test-and-set keeps a tile
            // from being listed twice, which would render it twice).
            if (cost >= Q.heavy_threshold) {
                const bool spread = cost >= RM_HEAVY_DILATE * Q.heavy_threshold;
                bool add = false;
                unsigned nt = tile;
                // the tile's column and row again, by the same multiplications as before the render (kept in registers
                // across the march they would cost more than this rare recomputation)
                unsigned base = 0, t2 = tile;
                if (SHAPE == 2 || (!PLAIN && U.n_poses != 1)) {
                    const unsigned per_pose = Q.tiles_x * Q.tiles_y;
                    base = (tile / per_pose) * per_pose; t2 = tile - base;
                }
                const int ty = (int)(unsigned)(((unsigned long long)t2 * Q.tx_magic) >> 37), tx = (int)(t2 - (unsigned)ty * Q.tiles_x);
                if (lane < 9 && (spread || lane == 4)) {
                    const int nx = tx + lane % 3 - 1, ny = ty + lane / 3 - 1;
                    if (nx >= 0 && ny >= 0 && nx < (int)Q.tiles_x && ny < (int)Q.tiles_y) {
                        nt = base + (unsigned)ny * Q.tiles_x + (unsigned)nx;
                        const unsigned bit = 1u << (nt & 31);
                        add = !(atomicOr(&bitmap_nxt[nt >> 5], bit) & bit);
                    }
                }
                const unsigned long long am = __ballot(add);
                if (am) {
                    unsigned idx = 0;
                    if (lane == 0) idx = atomicAdd(&Q.heavy_count[Q.hv_nxt], (unsigned)__builtin_popcountll(am));
                    idx = __builtin_amdgcn_readfirstlane(idx) + (unsigned)__builtin_popcountll(am & ((1ull << lane) - 1ull));
                    if (add) {
                        if (idx < Q.heavy_cap) heavy_nxt[idx] = nt;
                        else atomicAnd(&bitmap_nxt[nt >> 5], ~(1u << (nt & 31)));      // no room on the list: it stays in image order
                    }
                }
            }
#else
            if (cost >= Q.heavy_threshold && lane == 0) {
                unsigned idx = atomicAdd(&Q.heavy_count[Q.hv_nxt], 1u);
                if (idx < Q.heavy_cap) { heavy_nxt[idx] = tile; atomicOr(&bitmap_nxt[tile >> 5], 1u << (tile & 31)); }
            }
#endif
        }
        cur = nxt;
    }
    __builtin_amdgcn_s_setprio(0);

    c_hit = rm_wave_sum(c_hit); c_p = rm_wave_sum(c_p); c_s = rm_wave_sum(c_s);
    if (lane == 0) {
        unsigned *o = U.partials + RM_SLOT_WORDS * gw;
        o[0] = c_hit; o[1] = c_p; o[2] = c_s;
#ifdef RM_TIMELINE
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        o[3] = (unsigned)t_start; o[4] = (unsigned)(t_start >> 32); o[5] = (unsigned)t_end; o[6] = (unsigned)(t_end >> 32);
        o[7] = n_done | (max_cost << 16);
        o[0] = t_wait; o[1] = n_claims;       // timeline builds repurpose the counters
        for (int k = 0; k < 4; k++) { o[8 + 2 * k] = tl_t[k]; o[9 + 2 * k] = tl_c[k]; }
#endif
    }
}

// ---- kernel C: persistent waves with lane-level ray compaction ---------------------------
// Every lane carries one ray through a small state machine; one loop iteration is one scene
// evaluation for every busy lane, whatever that lane is doing: a primary step, one of the six
// normal taps (evaluated as o + d*t with d = +-axis, t = epsilon, which is hp +- e exactly) or
// a shadow step.  A lane whose pixel is finished goes idle and is refilled with the next pixel
// of the wave's current 8x8 tile as soon as RM_S_REFILL_MIN lanes are idle (a ballot), so a
// wave stays full while single rays of it run long.  The per-ray arithmetic is the same as in
// the tile kernels; only the interleaving differs, so the image is bit-identical.
//
// The colour evaluation (scene().rgb, a full blend tree) is the one step that is not a plain
// distance evaluation: lanes reaching it wait until RM_S_SHADE_MIN of them can share one
// divergent pass (or nothing else is left to do).
//
// Pays where one evaluation is expensive and step counts diverge (N-primitive scenes): the
// state handling is a few dozen instructions per iteration.  Tiles are claimed dynamically
// (first tile = wave index, then an atomic counter): a tile is milliseconds of work there.
#ifndef RM_S_REFILL_MIN
#define RM_S_REFILL_MIN 8
#endif
#ifndef RM_S_SHADE_MIN
#define RM_S_SHADE_MIN 8
#endif
enum { RM_PH_IDLE = 0, RM_PH_PRIMARY = 1, RM_PH_TAP = 2, RM_PH_WAITCOL = 3, RM_PH_SHADOW = 4 };

// COOP: the four waves of a workgroup carry the SAME 64 rays (the scene object splits every evaluation between them,
// rm_coop.h): the workgroup is the unit that owns a tile, its waves take every branch together, wave 0 writes.
template <bool COOP = false, class S>
RM_DI void rm_stream_body(const RmParams &U, const S &sc) {
    const RmQueue &Q = U.queue;
    const int lane = threadIdx.x & 63;
    const bool writer = !COOP || (threadIdx.x >> 6) == 0;
    __shared__ unsigned s_claim[2];
    unsigned claim_parity = 0;
    const unsigned gw = COOP ? blockIdx.x : __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const unsigned n_waves = COOP ? gridDim.x : (gridDim.x * blockDim.x) >> 6;
    unsigned *counter = Q.counters + Q.cnt_cur * RM_Q_SHARDS * RM_Q_STRIDE;
    if (Q.cnt_clr != RM_Q_SOLO && gw == 0 && writer && lane < RM_Q_SHARDS) Q.counters[(Q.cnt_clr * RM_Q_SHARDS + lane) * RM_Q_STRIDE] = 0u;

    const float sf = rm_vgpr(U.step_factor), kc = rm_vgpr(.0025f), eps = rm_vgpr(U.nss);
    const V3 L = mk3(rm_vgpr(U.light[0]), rm_vgpr(U.light[1]), rm_vgpr(U.light[2]));
    const unsigned per_pose = Q.tiles_x * Q.tiles_y;

    // per-lane ray state
    int phase = RM_PH_IDLE, cnt = 0, tap = 0;
    V3 o = mk3(0, 0, 0), d = mk3(0, 0, 0), raw = mk3(0, 0, 0), base = mk3(0, 0, 0);
    float t = 0.0f, lim = 0.0f, res = 1.0f, contact = 0.0f, tmp = 0.0f, ao = 0.0f, length = 0.0f;
    size_t gidx = 0, oidx = 0;
    unsigned c_hit = 0, c_p = 0, c_s = 0;

    unsigned tile = gw < Q.n_tiles ? gw : 0xffffffffu, next_px = 0;

    for (;;) {
        // ---- refill idle lanes from the current tile ------------------------------------
        const unsigned long long idle = __ballot(phase == RM_PH_IDLE);
        if (tile != 0xffffffffu && (__popcll(idle) >= RM_S_REFILL_MIN || idle == ~0ull)) {
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
            const unsigned px = next_px + rank;
            if (phase == RM_PH_IDLE && px < 64u) {
                const unsigned pose = tile / per_pose, t2 = tile - pose * per_pose;
                const unsigned ty = t2 / Q.tiles_x, tx = t2 - ty * Q.tiles_x;
                const int x = (int)(tx * 8u + (px & 7u)), yl = (int)(ty * 8u + (px >> 3));
                if (x < U.w && yl < U.rows) {
                    rm_setup_ray(U, x, yl, (int)pose, o, d, lim, gidx, oidx);
                    t = 0.0f; cnt = U.n_steps; contact = 0.0f;
                    phase = RM_PH_PRIMARY;
                    if (cnt <= 0) {                       // _MaxSteps <= 0: the loop body never runs, a miss of length 0
                        length = 0.0f;
                        phase = RM_PH_IDLE;
                        PixelOut po; po.oidx = oidx; po.depth = 0.0f; po.hit = 0;
                        po.rgba = U.src ? (reinterpret_cast<const uint32_t *>(U.src)[gidx] | 0xFF000000u) : U.bg_rgba8;
                        if (writer) rm_store_pixel(U, po);
                    }
                }
            }
            unsigned taken = (unsigned)__popcll(idle);
            taken = taken < 64u - next_px ? taken : 64u - next_px;
            next_px += taken;
            if (next_px == 64u) {
                unsigned kq = 0;
                if (COOP) {
                    // one claim per workgroup, handed to its waves through LDS; the slot alternates, so one barrier is enough
                    if (threadIdx.x == 0) s_claim[claim_parity] = atomicAdd(counter, 1u);
                    __syncthreads();
                    kq = s_claim[claim_parity];
                    claim_parity ^= 1u;
                } else {
                    if (lane == 0) kq = atomicAdd(counter, 1u);
                }
                kq = __builtin_amdgcn_readfirstlane(kq);
                tile = n_waves + kq;
                if (tile >= Q.n_tiles) tile = 0xffffffffu;
                next_px = 0;
            }
        }

        const bool marching = phase == RM_PH_PRIMARY || phase == RM_PH_TAP || phase == RM_PH_SHADOW;
        const unsigned long long m_mask = __ballot(marching);
        const unsigned long long w_mask = __ballot(phase == RM_PH_WAITCOL);
        if (!m_mask && !w_mask) {
            if (tile == 0xffffffffu) break;
            continue;                                     // only idle lanes: refill again
        }

        // ---- one scene evaluation for every marching lane --------------------------------
        if (marching) {
            const float dts = sc.w(mk3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t));
            if (phase == RM_PH_TAP) {
                // taps come in +/- pairs: x+, x-, y+, y-, z+, z-  (shader:63-67)
                if (!(tap & 1)) tmp = dts;
                else {
                    const float df = tmp - dts;
                    raw.x = tap == 1 ? df : raw.x; raw.y = tap == 3 ? df : raw.y; raw.z = tap == 5 ? df : raw.z;
                }
                tap++;
                if (tap == 6) phase = RM_PH_WAITCOL;
                else {
                    const float sg = (tap & 1) ? -1.0f : 1.0f;
                    d = mk3((tap >> 1) == 0 ? sg : 0.0f, (tap >> 1) == 1 ? sg : 0.0f, (tap >> 1) == 2 ? sg : 0.0f);
                }
            } else {
                if (phase == RM_PH_SHADOW) { res = fminf(res, 16.0f * dts / t); c_s++; }      // shader:111
                else c_p++;
                const float a = fabsf(dts);
                t += a * sf;
                contact = t * kc;
                cnt--;
                const bool touch = a < contact;
                if (touch || t > lim || cnt == 0) {
                    if (phase == RM_PH_PRIMARY) {
                        length = t;
                        if (touch && U.mode == RM_MODE_FRAG) {
                            // hit: on to the normal taps around hp = ro + rd * r.length (shader:174)
                            c_hit++;
                            o = mk3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
                            d = mk3(1.0f, 0.0f, 0.0f);
                            t = eps; tap = 0;
                            phase = RM_PH_TAP;
                        } else {
                            if (touch) c_hit++;
                            PixelOut po; po.oidx = oidx; po.depth = t; po.hit = touch ? 1u : 0u;
                            po.rgba = U.src ? (reinterpret_cast<const uint32_t *>(U.src)[gidx] | 0xFF000000u) : U.bg_rgba8;
                            if (writer) rm_store_pixel(U, po);
                            phase = RM_PH_IDLE;
                        }
                    } else {
                        // shadow march finished: shade and write (shader:195-199)
                        const float shadow = res * 0.75f + .25f;
                        PixelOut po; po.oidx = oidx; po.depth = length; po.hit = 1u;
                        po.rgba = rm_unorm8((base.x * shadow) * ao) | (rm_unorm8((base.y * shadow) * ao) << 8) |
                                  (rm_unorm8((base.z * shadow) * ao) << 16) | 0xFF000000u;
                        if (writer) rm_store_pixel(U, po);
                        phase = RM_PH_IDLE;
                    }
                }
            }
        }

        // ---- colour + shading set-up for lanes that finished their taps, in batches ----------
        const unsigned long long w2 = __ballot(phase == RM_PH_WAITCOL);
        const unsigned long long m2 = __ballot(phase == RM_PH_PRIMARY || phase == RM_PH_TAP || phase == RM_PH_SHADOW);
        if (w2 && (__popcll(w2) >= RM_S_SHADE_MIN || (!m2 && (tile == 0xffffffffu || __popcll(__ballot(phase == RM_PH_IDLE)) < RM_S_REFILL_MIN)))) {
            if (phase == RM_PH_WAITCOL) {
                const V3 n = rm_normalize(raw);
                const V4 s = sc.full(o);                                                          // shader:193, o == hitpoint
                const float ndl = rm_dot3(n, L) * .5f + .5f;
                ao = rm_clamp01(rm_len3(mk3(raw.x / U.nss_raw, raw.y / U.nss_raw, raw.z / U.nss_raw)));   // shader:125-127
                base = mk3(s.x * ndl, s.y * ndl, s.z * ndl);
                // getshadow: start at p + n * _ContactThreshold * 2 with the threshold raymarch() left (shader:107)
                const float c0 = length * kc;
                o = mk3(o.x + (n.x * c0) * 2.0f, o.y + (n.y * c0) * 2.0f, o.z + (n.z * c0) * 2.0f);
                d = L;
                t = 0.0f; res = 1.0f; lim = U.shadow_depth; cnt = U.n_steps;
                phase = RM_PH_SHADOW;
            }
        }
    }

    c_hit = rm_wave_sum(c_hit); c_p = rm_wave_sum(c_p); c_s = rm_wave_sum(c_s);
    if (COOP && !writer) c_hit = c_p = c_s = 0;        // the four waves counted the same rays: one slot per wave, three of them empty
    if (lane == 0) {
        const unsigned slot = COOP ? blockIdx.x * 4u + (threadIdx.x >> 6) : gw;
        unsigned *out = U.partials + RM_SLOT_WORDS * slot;
        out[0] = c_hit; out[1] = c_p; out[2] = c_s;
#ifdef RM_TIMELINE
        out[3] = out[4] = out[5] = out[6] = out[7] = 0;
#endif
    }
}

// The N-primitive table is staged into LDS once per workgroup (96 B per primitive, up to
// RM_LDS_PRIMS of them): every lane then reads the same record, which the LDS serves as a
// broadcast into VGPRs -- no scalar-load latency per primitive and no SGPR operands in the
// distance arithmetic.  Larger tables are read from global memory through the scalar cache.
RM_DI RmLdsPrimPtr rm_stage_table(const RmParams &U, unsigned lds_word_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned s_dyn[];
    unsigned *base = s_dyn + lds_word_offset;
    const uint4 *src = reinterpret_cast<const uint4 *>(U.prim_table);
    uint4 *dst = reinterpret_cast<uint4 *>(base);
    const unsigned n16 = (unsigned)U.lds_prims * (sizeof(RmPrimDev) / 16);
    for (unsigned i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    return (RmLdsPrimPtr)(base);
}

// Waves per SIMD the register allocator is asked to fit.  4 (<= 128 VGPRs) is the measured best
// for the persistent kernel on the reference's own scene (3 starves the VALU, 5+ slows the waves
// that carry a frame's longest tiles).  Variants that divide by a non-unit scale keep ~30 more
// constants live and get 3 waves (<= 168 VGPRs); variants with per-primitive rotation keep ~50
// more and get 2 (<= 256): any spill lands inside the march loop and costs far more than the
// occupancy (rotated + scaled 4-slot scene at 1080p: 700 us with spills at 3 waves, 258 us at 2).
#ifndef RM_Q_MIN_WAVES
#define RM_Q_MIN_WAVES 4
#endif
#ifndef RM_Q_MIN_WAVES_SCL
#define RM_Q_MIN_WAVES_SCL 3
#endif
#ifndef RM_Q_MIN_WAVES_ROT
#define RM_Q_MIN_WAVES_ROT 2
#endif
#define RM_QLB __launch_bounds__(256, (ROT ? RM_Q_MIN_WAVES_ROT : (SCL == RM_SCL_UNIT ? RM_Q_MIN_WAVES : RM_Q_MIN_WAVES_SCL)))

// kernel variant of a scene: rotation yes/no, scale unit / reciprocal-exact / true division; index 0..5
inline int rm_scene_variant(const RmParams &U, bool &rot, int &scl) {
    rot = !(U.scene_flags & RM_SF_IDENT_ROT);
    scl = (U.scene_flags & RM_SF_UNIT_SCALE) ? RM_SCL_UNIT : (U.scene_flags & RM_SF_RCP) ? RM_SCL_RCP : RM_SCL_DIV;
    return (rot ? 3 : 0) + scl;
}

// calls F<ROT, SCL>::run(args...) for the variant the scene needs
#define RM_DISPATCH_VARIANT(F, U, ...)                                                   \
    do {                                                                                 \
        bool rot_; int scl_;                                                             \
        rm_scene_variant(U, rot_, scl_);                                                 \
        if (rot_) {                                                                      \
            if (scl_ == RM_SCL_UNIT) return F<true, RM_SCL_UNIT>(__VA_ARGS__);           \
            if (scl_ == RM_SCL_RCP)  return F<true, RM_SCL_RCP>(__VA_ARGS__);            \
            return F<true, RM_SCL_DIV>(__VA_ARGS__);                                     \
        }                                                                                \
        if (scl_ == RM_SCL_UNIT) return F<false, RM_SCL_UNIT>(__VA_ARGS__);              \
        if (scl_ == RM_SCL_RCP)  return F<false, RM_SCL_RCP>(__VA_ARGS__);               \
        return F<false, RM_SCL_DIV>(__VA_ARGS__);                                        \
    } while (0)

}  // namespace
