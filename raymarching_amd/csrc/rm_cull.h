#pragma once
// rm_cull.h -- the tolerance kernel for N-primitive smooth-union scenes: every march step evaluates only the primitives
// that can reach the result.
//
// A leaf of the balanced smin tree (Assets/Raymarching.cginc:110-113, 363, generalised to N leaves) with distance d
// cannot change the root if d >= root + k * depth: smin(a, b) returns a exactly once b >= a + k, it never returns less
// than min(a, b) - k/4, and it is monotone, so a change at a leaf that far above the root value is clamped away at
// some node on the way up (by induction over the levels: a node whose value differs has a value above
// root + k * (levels left)).  With k = .5 and 256 leaves that is a slack of 4.  The kernel knows an upper bound of the
// root at the next sample point before it evaluates it -- the scene function is 1-Lipschitz, so
// f(q) <= |f(p)| + |q - p| -- and a lower bound of every leaf from a bounding sphere:
// leaf_i(q) >= (|q - c_i| - R_i) * min(scale_i) / max(scale_i).  One wave marches the 64 rays of an 8x8 tile; after
// every step it takes the bounding box of its rays' next sample points, tests all primitives against it cooperatively
// (one primitive per lane) and keeps the survivors as a bit mask in scalar registers.  The next evaluation walks
// the set bits in index order and rebuilds the same tree over the leaves that are present (an absent sibling passes
// its partner up unchanged), so mathematically the value is the full tree's; in floating point it differs in the last
// bits (b + 1*(a - b) is not a), which is why this lives in the tolerance kernel.  `_RmCullSlack` overrides the slack
// (smaller = faster, no longer covered by the argument above).  The normal taps and the colour evaluation walk the
// whole table with the strict arithmetic.
#include "rm_kernels_common.h"
#include "rm_fast.h"

namespace rmc {

typedef float rm_f4v __attribute__((ext_vector_type(4)));

// wave-wide float min / max with all 64 lanes active; the result is in every lane
template <bool MAX> RM_DI float wave_reduce(float v) {
    float t;
#define RM_RSTEP(ctrl)                                                                                             \
    t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true));                  \
    v = MAX ? fmaxf(v, t) : fminf(v, t);
    RM_RSTEP(0xB1) RM_RSTEP(0x4E) RM_RSTEP(0x141) RM_RSTEP(0x140)
#undef RM_RSTEP
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : fminf(fminf(a, b), fminf(c, d));
}

#define RM_CULL_WORDS 8          // 512 primitives at most (the LDS table's limit)
#define RM_CULL_LEVELS 10

// FASTM: the leaf and node arithmetic of rm_fast.h (fused multiply-adds, raw sqrt, reciprocal scale); otherwise the strict
// arithmetic of rm_math.h, so that culling is the only difference from the exact kernels
template <bool ROT, bool SCALED, bool FASTM>
struct SceneNCull {
    static constexpr bool fastm = FASTM;
    float tiny;
    PrimTable<true> tab;
    int n, n_words;
    const RM_LDS rm_f4v *cr;     // per primitive: bounding sphere (centre, radius) in world space
    const RM_LDS float *ir;      // per primitive: max(scale) / min(scale)
    float slack;
    unsigned long long mask[RM_CULL_WORDS];       // wave-uniform: the primitives the next evaluation visits

    RM_DI void mask_all() {
#pragma unroll
        for (int w = 0; w < RM_CULL_WORDS; w++) {
            const int left = n - 64 * w;
            mask[w] = left >= 64 ? ~0ull : (left > 0 ? ((1ull << left) - 1ull) : 0ull);
        }
    }

    // Every lane calls this (full exec).  `active` lanes will evaluate the scene at q next; thr = upper bound of the scene
    // value at q plus the slack.
    RM_DI void build_mask(bool active, V3 q, float thr) {
        const float inf = __builtin_inff();
        if (!__any(active)) return;
        const float lox = wave_reduce<false>(active ? q.x : inf), hix = wave_reduce<true>(active ? q.x : -inf);
        const float loy = wave_reduce<false>(active ? q.y : inf), hiy = wave_reduce<true>(active ? q.y : -inf);
        const float loz = wave_reduce<false>(active ? q.z : inf), hiz = wave_reduce<true>(active ? q.z : -inf);
        const float thrw = wave_reduce<true>(active ? thr : -inf) + 1e-3f;
        if (!(thrw < 1e30f) || !(hix - lox < 1e30f) || !(hiy - loy < 1e30f) || !(hiz - loz < 1e30f)) { mask_all(); return; }   // NaN / inf: no culling
        const float cx = 0.5f * (lox + hix), cy = 0.5f * (loy + hiy), cz = 0.5f * (loz + hiz);
        const float ex = hix - lox, ey = hiy - loy, ez = hiz - loz;
        // half diagonal of the box, rounded up a little: every active q is within hd of (cx, cy, cz)
        const float hd = 0.5f * __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + 1e-5f * (fabsf(cx) + fabsf(cy) + fabsf(cz));
        const int lane = threadIdx.x & 63;
        unsigned long long any_bits = 0;
#pragma unroll
        for (int w = 0; w < RM_CULL_WORDS; w++) {
            if (w < n_words) {
                const int i = 64 * w + lane;
                bool keep = false;
                if (i < n) {
                    const rm_f4v c = cr[i];
                    const float dx = c.x - cx, dy = c.y - cy, dz = c.z - cz;
                    const float dist = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
                    keep = (dist - hd - c.w) < thrw * ir[i];
                }
                mask[w] = __ballot(keep);
                any_bits |= mask[w];
            } else {
                mask[w] = 0ull;
            }
        }
        if (any_bits == 0ull) mask_all();
    }

    // the smin tree over the primitives whose mask bit is set, in the shape the full tree has
    RM_DI float eval(V3 p) const {
        float cur = 1e30f;
        unsigned cur_idx = 0, pendmask = 0;
        bool have = false;
        float pend[RM_CULL_LEVELS + 1];
#pragma unroll
        for (int l = 0; l <= RM_CULL_LEVELS; l++) pend[l] = 0.0f;
#pragma unroll
        for (int w = 0; w < RM_CULL_WORDS; w++) {
            if (w >= n_words) break;
            unsigned long long m = mask[w];
            while (m) {
                const unsigned b = (unsigned)__builtin_ctzll(m);
                m &= m - 1ull;
                const unsigned i = 64u * (unsigned)w + b;
                const auto P = tab.get((int)i);
                const float d = FASTM ? rmf::prim<ROT, SCALED>(P, P.type & RM_TYPE_MASK, p)
                                      : rm_prim_typed<ROT, SCALED ? RM_SCL_RCP : RM_SCL_UNIT>(P, P.type & RM_TYPE_MASK, p, tiny);
                if (have) {
                    // level of the node where the subtree ending at cur_idx and leaf i meet: everything pending below it closes first
                    const unsigned lvl = 32u - (unsigned)__builtin_clz(cur_idx ^ i);
#pragma unroll
                    for (unsigned l = 1; l <= RM_CULL_LEVELS; l++)
                        if (((pendmask >> l) & 1u) && l < lvl) cur = FASTM ? rmf::smin_w(pend[l], cur) : rm_smin_w(pend[l], cur);
                    pendmask &= ~((1u << lvl) - 1u);
#pragma unroll
                    for (unsigned l = 1; l <= RM_CULL_LEVELS; l++)
                        if (l == lvl) pend[l] = cur;
                    pendmask |= 1u << lvl;
                }
                cur = d; cur_idx = i; have = true;
            }
        }
#pragma unroll
        for (unsigned l = 1; l <= RM_CULL_LEVELS; l++)
            if ((pendmask >> l) & 1u) cur = FASTM ? rmf::smin_w(pend[l], cur) : rm_smin_w(pend[l], cur);
        return cur;
    }
};

// The fragment program for one pixel with the two marches on the culled evaluation.  Every lane of the wave calls this
// (lanes outside the frame with in_bounds = false): the mask is built by the whole wave.
template <class CULL, class STRICT>
RM_DI PixelOut compute_pixel(const RmParams &U, CULL &sc, const STRICT &strict, int x, int yl, int pose, bool in_bounds,
                             unsigned &c_hit, unsigned &c_p, unsigned &c_s) {
    PixelOut out;
    V3 ro, rd;
    float depth;
    size_t gidx;
    rm_setup_ray(U, x, yl, pose, ro, rd, depth, gidx, out.oidx);
    constexpr bool FM = CULL::fastm;
    const float sf = rm_vgpr(U.step_factor), kc = rm_vgpr(.0025f);
    const V3 L = mk3(rm_vgpr(U.light[0]), rm_vgpr(U.light[1]), rm_vgpr(U.light[2]));

    // ---- raymarch (cginc:367-392) ----
    sc.mask_all();
    bool active = in_bounds && U.n_steps > 0;
    float dm = 0.0f, adts = __builtin_inff(), contact = 0.0f;
    unsigned evals = 0;
    const float rdlen = __builtin_amdgcn_sqrtf(rm_dot3(rd, rd)) * 1.0001f;
    for (int i = 0; i < U.n_steps; i++) {
        if (!__any(active)) break;
        const V3 cp = FM ? mk3(__builtin_fmaf(rd.x, dm, ro.x), __builtin_fmaf(rd.y, dm, ro.y), __builtin_fmaf(rd.z, dm, ro.z))
                         : mk3(ro.x + rd.x * dm, ro.y + rd.y * dm, ro.z + rd.z * dm);
        const float v = sc.eval(cp);
        if (active) {
            adts = fabsf(v);
            evals++;
            dm = FM ? __builtin_fmaf(adts, sf, dm) : dm + adts * sf;
            contact = dm * kc;
            if (adts < contact || dm > depth) active = false;
        }
        const V3 q = mk3(__builtin_fmaf(rd.x, dm, ro.x), __builtin_fmaf(rd.y, dm, ro.y), __builtin_fmaf(rd.z, dm, ro.z));
        sc.build_mask(active, q, adts + (adts * sf) * rdlen + sc.slack);
    }
    const bool hit = in_bounds && adts < contact;
    c_p += evals;
    c_hit += hit ? 1u : 0u;
    out.depth = dm;
    out.hit = hit ? 1u : 0u;
    out.rgba = 0;
    out.steps = 0;

    if (U.mode == RM_MODE_FRAG) {
        // ---- normal, colour (strict, whole table) ----
        V3 hp = mk3(ro.x + rd.x * dm, ro.y + rd.y * dm, ro.z + rd.z * dm), n = mk3(0.0f, 0.0f, 0.0f), raw = n;
        V4 s;
        s.x = s.y = s.z = s.w = 0.0f;
        if (hit) {
            const float e = U.nss;
#pragma nounroll
            for (int a = 0; a < 3; a++) {
                const V3 o = mk3(a == 0 ? e : 0.0f, a == 1 ? e : 0.0f, a == 2 ? e : 0.0f);
                const float d = strict.w(mk3(hp.x + o.x, hp.y + o.y, hp.z + o.z)) - strict.w(mk3(hp.x - o.x, hp.y - o.y, hp.z - o.z));
                raw.x = a == 0 ? d : raw.x; raw.y = a == 1 ? d : raw.y; raw.z = a == 2 ? d : raw.z;
            }
            n = rm_normalize(raw);
            s = strict.full(hp);
        }
        // ---- getshadow (shader:102-123), hit lanes only ----
        const float c0 = contact;
        const V3 p = mk3(hp.x + (n.x * c0) * 2.0f, hp.y + (n.y * c0) * 2.0f, hp.z + (n.z * c0) * 2.0f);
        const float Llen = __builtin_amdgcn_sqrtf(rm_dot3(L, L)) * 1.0001f;
        bool sactive = hit && U.n_steps > 0;
        // the start point is within one primary step plus the offset of the last primary sample
        sc.build_mask(sactive, p, adts + (adts * sf) * rdlen + 2.0f * c0 + sc.slack);
        float sdm = 0.0f, res = 1.0f;
        unsigned sev = 0;
        for (int i = 0; i < U.n_steps; i++) {
            if (!__any(sactive)) break;
            const V3 cp = FM ? mk3(__builtin_fmaf(L.x, sdm, p.x), __builtin_fmaf(L.y, sdm, p.y), __builtin_fmaf(L.z, sdm, p.z))
                             : mk3(p.x + L.x * sdm, p.y + L.y * sdm, p.z + L.z * sdm);
            const float dts = sc.eval(cp);
            float a = 0.0f;
            if (sactive) {
                sev++;
                res = fminf(res, FM ? (16.0f * dts) * __builtin_amdgcn_rcpf(sdm) : 16.0f * dts / sdm);
                a = fabsf(dts);
                sdm = FM ? __builtin_fmaf(a, sf, sdm) : sdm + a * sf;
                const float ct = sdm * kc;
                if (a < ct || sdm > U.shadow_depth) sactive = false;
            }
            const V3 q = mk3(__builtin_fmaf(L.x, sdm, p.x), __builtin_fmaf(L.y, sdm, p.y), __builtin_fmaf(L.z, sdm, p.z));
            sc.build_mask(sactive, q, a + (a * sf) * Llen + sc.slack);
        }
        c_s += sev;
        if (hit) {
            const float shadow = res * 0.75f + .25f;
            const float ndl = rm_dot3(n, L) * .5f + .5f;
            const float ao = rm_clamp01(rm_len3(mk3(raw.x / U.nss_raw, raw.y / U.nss_raw, raw.z / U.nss_raw)));
            out.rgba = rm_unorm8(((s.x * ndl) * shadow) * ao) | (rm_unorm8(((s.y * ndl) * shadow) * ao) << 8) |
                       (rm_unorm8(((s.z * ndl) * shadow) * ao) << 16) | 0xFF000000u;
        } else if (in_bounds) {
            out.rgba = U.src ? (reinterpret_cast<const uint32_t *>(U.src)[gidx] | 0xFF000000u) : U.bg_rgba8;
        }
    }
    return out;
}

}  // namespace rmc
