#pragma once
// rm_skip.h -- exact culling for table scenes in the shared-tile kernel: a march step evaluates only the
// primitives that can reach the result, and the result keeps every bit.
//
// smin(a, b, .5) (Assets/Raymarching.cginc:110-113) is h = clamp(.5 + (b - a), 0, 1); lerp(b, a, h) - .5*h*(1-h), evaluated
// here as e = b - a; (b - h*e) - (.5*h)*(1 - h) (rm_smin_w).  Once b <= a - .5 the sum .5 + e rounds to at most 0, the
// clamp gives h = 0 and the value is (b - 0*e) - 0*1: the RIGHT operand, bit for bit, whatever finite value the left operand
// has.  (b = -0 would come back as +0; a leaf of a plain union with positive scales never returns -0 -- its last operation
// is a difference or a sum of values that are not -0, times a positive factor -- and a node returns -0 only if b is.)
// The other side is not exact: b - 1*(b - a) is not a in floating point.  So a node's LEFT subtree need not be evaluated if
// its value is known to be at least .5 above the right subtree's.
//
// The bounds: every leaf rotate(p - pos, q) / scale -> primitive -> * min(scale) (cginc:336-354) is 1-Lipschitz in p
// (the primitives are Euclidean distance functions; |grad| <= min(scale) / min(scale), scales positive), and
// min(a, b) - .125 <= smin(a, b) <= min(a, b).  Before an evaluation the wave takes the bounding box of its marching rays'
// sample points and evaluates the 64 leaves of its chunk at the box centre, one leaf per lane: that value minus / plus the
// half diagonal bounds the leaf on the whole box from below / above.  A butterfly of minima gives both bounds for every
// aligned block of 2^l leaves (the lower one less .126 per level), and a left block is skipped when its sibling's upper
// bound is .5 below its lower bound.  A margin of .01 covers the rounding of the fp32 evaluations with all coordinates
// and distances below 1000; beyond that, or with anything not finite, nothing is skipped.  The decisions are taken before
// the walk (as a bit mask over the leaves), so the walk is SceneN::tree's with absent operands: straight-line groups of four
// with their loads issued together, a binary-counter stack over the groups.  The last leaf of a chunk is always evaluated.
#include "rm_kernels_common.h"
#include "rm_coop.h"

namespace rmx {

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

#define RM_SKIP_LEVELS 6          // levels of the stack over groups of four leaves: chunks of up to 128 leaves
#define RM_SKIP_MAX_CHUNK 128     // two leaves per lane in the bounds, two mask words
#ifndef RM_SKIP_MIN_CHUNK
#define RM_SKIP_MIN_CHUNK 16      // quarters of 16 leaves (33 to 64 primitives) still gain 3-14 %
#endif
#define RM_SKIP_RANGE 1000.0f     // coordinates and distances beyond this: no culling (the margin below is sized for it)

// the record of leaf i as this lane's own (lanes read different records; PrimTable::get is for wave-uniform indices)
RM_DI RmPrimDev lane_record(RmLdsPrimPtr T, int i) {
    union { RmPrimDev p; rm_u4 v[sizeof(RmPrimDev) / 16]; } u;
#pragma unroll
    for (int j = 0; j < (int)(sizeof(RmPrimDev) / 16); j++) u.v[j] = T[i * (int)(sizeof(RmPrimDev) / 16) + j];
    return u.p;
}

// BIG: quarters of 128 leaves (tables of 257 to 512 primitives); kept apart so that the smaller tables do not carry the
// second half's code (the 256-primitive stress scene was 2.7 % slower with both in one kernel)
// CSG: the table carries op bits.  An smax node (cginc:120-123 as SceneN::comb evaluates it: h = clamp(.5 + (a - b)))
// returns its right operand bit for bit once the LEFT one is .5 BELOW it, and lies in [max(a, b) - 1/6, max(a, b)]; a
// negated leaf swaps its bounds.  A skipped left operand leaves b - (+0) = b under an smax node and b - (-0) = b + 0 under
// an smin node, which differ for b = -0 (a negated leaf can return it).
template <bool ROT, int SCL, bool BIG, bool CSG>
struct SceneNSkip {
    const SceneNCoop<ROT, SCL> &coop;
    bool cullable;                 // this wave's chunk is complete and has RM_SKIP_MIN_CHUNK..RM_SKIP_MAX_CHUNK leaves
    int groups;                    // leaves of the chunk / 4, a power of two
    unsigned long long all[2];     // one bit per leaf of the chunk
    unsigned long long mask[2];    // the leaves the next evaluation visits
    mutable unsigned n_visited = 0, n_possible = 0;    // leaf evaluations this wave executed / a full walk of its chunk would have
#ifdef RM_SKIP_STATS               // developer build: leaves visited by the slowest wave / by all four, per evaluation
    mutable unsigned st_max = 0, st_sum = 0, st_n = 0, st_par = 0;
    int st_phase = 0;              // 1 primary march, 2 shadow march
    RM_LDS unsigned *st_lds = nullptr;
#endif

    RM_DI SceneNSkip(const SceneNCoop<ROT, SCL> &c, int chunk) : coop(c) {
        cullable = c.part.n == chunk && chunk >= (CSG ? 2 * RM_SKIP_MIN_CHUNK : RM_SKIP_MIN_CHUNK) && chunk <= (BIG ? RM_SKIP_MAX_CHUNK : 64) && (!BIG || chunk > 64);
        groups = chunk >> 2;
        all[0] = chunk >= 64 ? ~0ull : ((1ull << chunk) - 1ull);
        all[1] = chunk > 64 ? ~0ull : 0ull;
        mask[0] = all[0]; mask[1] = all[1];
    }

    // Every lane of the wave calls this; `active` lanes evaluate the scene next, at q or anywhere within `grow` of q along
    // each axis.
    RM_DI void prepare(bool active, V3 q, float grow = 0.0f) {
        mask[0] = all[0]; mask[1] = all[1];
        if (!cullable || !__any(active)) return;
        const float inf = __builtin_inff();
        const float lox = wave_reduce<false>(active ? q.x : inf), hix = wave_reduce<true>(active ? q.x : -inf);
        const float loy = wave_reduce<false>(active ? q.y : inf), hiy = wave_reduce<true>(active ? q.y : -inf);
        const float loz = wave_reduce<false>(active ? q.z : inf), hiz = wave_reduce<true>(active ? q.z : -inf);
        const float cx = 0.5f * (lox + hix), cy = 0.5f * (loy + hiy), cz = 0.5f * (loz + hiz);
        const float ex = hix - lox, ey = hiy - loy, ez = hiz - loz;
        // every marching sample point is within hd of the centre (rounded up), .01 for the rounding of the evaluations
        const float hd = 0.5f * __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + 1e-5f * (fabsf(cx) + fabsf(cy) + fabsf(cz)) + 0.01f +
                         1.7321f * grow;
        const int j = (int)coop.lane;
        const int n_leaves = 4 * groups;
        constexpr int halves = BIG ? 2 : 1;
        bool bad = !(fabsf(cx) + fabsf(cy) + fabsf(cz) + hd <= RM_SKIP_RANGE);          // also true for NaN
        unsigned long long keep[2] = {0ull, 0ull};
        float lb_half0 = inf, ub_half0 = inf;           // bounds of leaves 0..63 as a block (chunks of 128)
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            if (hf >= halves) break;           // (compile time)
            const bool mine = 64 * hf + j < n_leaves;
            float lb = inf, ub = inf, osg = 1.0f;
            if (mine) {
                const RmPrimDev P = lane_record(coop.part.tab.T, 64 * hf + j);
                float d = rm_prim_typed<ROT, SCL>(P, P.type & RM_TYPE_MASK, mk3(cx, cy, cz), coop.part.tiny);
                if (CSG) { d = d * P.nsgn; osg = P.osgn; }
                // (a negative scale would make min(scale) the largest in magnitude and the leaf steeper than 1)
                bad = bad || !(fabsf(d) + hd <= RM_SKIP_RANGE) || !(fabsf(P.px) + fabsf(P.py) + fabsf(P.pz) <= RM_SKIP_RANGE) ||
                      (SCL != RM_SCL_UNIT && !(P.ms > 0.0f));
                lb = d - hd;
                ub = d + hd;
            }
            // level l: the aligned blocks of 2^l leaves meet their siblings; a left block (bit l of j clear) is skipped if the
            // right one's upper bound is .5 below its lower bound, which loses .126 per level of smin inside it
            bool skip = false;
#pragma unroll
            for (int l = 0; l < 6; l++) {
                const float sub = __shfl_xor(ub, 1 << l), slb = __shfl_xor(lb, 1 << l);
                const bool left = !((j >> l) & 1);
                if (!CSG) {
                    skip = skip || (left && sub + (0.5f + 0.126f * (float)l) <= lb);
                    ub = fminf(ub, sub);
                    lb = fminf(lb, slb);
                } else {
                    // the node's op sits in the first leaf of its right operand; here the bounds carry their slack along
                    const bool smax = __shfl(osg, ((j >> l) | 1) << l) < 0.0f;
                    const float lbL = left ? lb : slb, ubL = left ? ub : sub, lbR = left ? slb : lb, ubR = left ? sub : ub;
                    skip = skip || (left && (smax ? ubL + 0.5f <= lbR : ubR + 0.5f <= lbL));
                    ub = smax ? fmaxf(ub, sub) : fminf(ub, sub);
                    lb = smax ? fmaxf(lb, slb) - 0.168f : fminf(lb, slb) - 0.126f;
                }
            }
            // (after the butterfly lb and ub are those of all 64 leaves of this half, in every lane)
            const unsigned long long k = __ballot(mine && !skip);
            if (hf == 0) {
                lb_half0 = lb; ub_half0 = ub;
                keep[0] = k;
            } else {
                keep[1] = k;
                // the first half as a block against the second
                if (!CSG) {
                    if (__all(ub + (0.5f + 0.126f * 6.0f) <= lb_half0)) keep[0] = 0ull;
                } else {
                    const bool smax = __shfl(osg, 0) < 0.0f;
                    if (__all(smax ? ub_half0 + 0.5f <= lb : ub + 0.5f <= lb_half0)) keep[0] = 0ull;
                }
            }
        }
        if (__any(bad)) return;
#ifndef RM_SKIP_NEVER
        mask[0] = keep[0]; mask[1] = keep[1];
#endif
    }

    RM_DI float leaf(int i, V3 p, float &osgn, int &tw) const { return coop.part.template leaf_w<CSG>(i, p, osgn, tw); }
    RM_DI static float comb(float a, float b, float s) { return SceneN<ROT, SCL, true>::template comb<CSG>(a, b, s); }
    // the value of a node whose left operand was skipped, s its op
    RM_DI static float alone(float b, float s) { return CSG ? b + (s > 0.0f ? 0.0f : -0.0f) : b; }
    // group g of four leaves a, b, c, d with the leaves m4 names (bit 3, the last leaf, is set whenever any is), as
    // comb(comb(a, b), comb(c, d)); tw receives the type word of the group's first record, which carries the ops of the
    // stack merges that follow.  The op of a node sits in the first leaf of its right operand, present or not.
    // Measured and dropped: fetching all four records up front whatever m4 says (no gain), one rolled loop over the present
    // leaves (18 % slower: the four leaves of a full group overlap in the pipeline), straight-line code per pattern of
    // present leaves (2 % slower).
    RM_DI float group(int g, unsigned m4, V3 p, int &tw) const {
        const int i = g << 2;
        float sa = 1.0f, sb = 1.0f, sc = 1.0f, sd = 1.0f;
        int t1 = 0;
        tw = 0;
        if (m4 == 15u) {
            const float a = leaf(i, p, sa, tw), b = leaf(i + 1, p, sb, t1), c = leaf(i + 2, p, sc, t1), d = leaf(i + 3, p, sd, t1);
            return comb(comb(a, b, sb), comb(c, d, sd), sc);
        }
        const float d = leaf(i + 3, p, sd, t1);
        float c = 0.0f;
        if (m4 & 4u) c = leaf(i + 2, p, sc, t1);
        else if (CSG) sc = coop.part.tab.get(i + 2).osgn;
        const float cd = (m4 & 4u) ? comb(c, d, sd) : alone(d, sd);
        if (!(m4 & 2u)) {
            if (CSG) tw = coop.part.tab.get(i).type;
            return alone(cd, sc);
        }
        const float b = leaf(i + 1, p, sb, t1);
        float a = 0.0f;
        if (m4 & 1u) a = leaf(i, p, sa, tw);
        else if (CSG) tw = coop.part.tab.get(i).type;
        const float ab = (m4 & 1u) ? comb(a, b, sb) : alone(b, sb);
        return comb(ab, cd, sc);
    }
    // this wave's chunk: the binary-counter walk of SceneN::tree over the groups that are present
    RM_DI float walk(V3 p) const {
        constexpr int LEVELS = BIG ? RM_SKIP_LEVELS : RM_SKIP_LEVELS - 1;
        float st[LEVELS];
#pragma unroll
        for (int l = 0; l < LEVELS; l++) st[l] = 0.0f;
        unsigned pres = 0;                                      // bit l: st[l] holds a subtree (else it was skipped)
        float v = 0.0f;
        for (int g = 0; g < groups; g++) {
            const unsigned m4 = BIG ? (unsigned)((g < 16 ? mask[0] : mask[1]) >> (4 * (g & 15))) & 15u : (unsigned)(mask[0] >> (4 * g)) & 15u;
            const int merges = __builtin_ctz(~(unsigned)g);     // trailing ones of g = finished lower levels
            const unsigned below = (1u << merges) - 1u;
            if (m4 == 0u) {                                     // a skipped block is a whole left operand: nothing below it is kept
                pres &= ~(below | (1u << merges));
                continue;
            }
            int tw;
            v = group(g, m4, p, tw);
#pragma unroll
            for (int l = 0; l < LEVELS - 1; l++) {
                if (l < merges) {
                    const float sl = (CSG && ((tw >> (RM_CASCADE_SHIFT + l)) & 1)) ? -1.0f : 1.0f;
                    if ((pres >> l) & 1u) v = comb(st[l], v, sl);
                    else if (CSG) v = alone(v, sl);
                }
            }
#pragma unroll
            for (int l = 0; l < LEVELS; l++) if (l == merges) st[l] = v;
            pres = (pres & ~below) | (1u << merges);
        }
        return v;
    }
    // scene distance at p for the marches; prepare() must have seen p
    RM_DI float eval(V3 p) const {
        float mine = 0.0f;
        if (cullable) mine = walk(p);
        else if (coop.part.n > 0) mine = coop.part.w(p);
        n_visited += cullable ? (unsigned)(__builtin_popcountll(mask[0]) + __builtin_popcountll(mask[1])) : (unsigned)coop.part.n;
        n_possible += (unsigned)coop.part.n;
#ifdef RM_SKIP_STATS
        if (coop.lane == 0) st_lds[st_par * 4u + coop.wv] = (unsigned)(__builtin_popcountll(mask[0]) + __builtin_popcountll(mask[1]));
        const float r = coop.finish(mine);                 // has the barrier
        if (st_phase == RM_SKIP_STATS) {
            const unsigned a = st_lds[st_par * 4u], b = st_lds[st_par * 4u + 1u], c = st_lds[st_par * 4u + 2u], d = st_lds[st_par * 4u + 3u];
            const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
            st_max += ab > cd ? ab : cd; st_sum += a + b + c + d; st_n++;
        }
        st_par ^= 1u;
        return r;
#else
        return coop.finish(mine);
#endif
    }
};

// the culled evaluation under the name the normal taps call
template <class SKIP> struct Taps {
    const SKIP &sc;
    RM_DI float w(V3 p) const { return sc.eval(p); }
};

// The fragment program for one pixel with the two marches and the six normal taps on the culled evaluation.  Every lane of the wave calls this
// (lanes outside the frame with in_bounds = false and the coordinates of some pixel inside): the bounds are built by the
// whole wave.  Operations and their order are those of rm_compute_pixel.
template <class SKIP, class STRICT>
RM_DI PixelOut compute_pixel(const RmParams &U, SKIP &sc, const STRICT &strict, int x, int yl, int pose, bool in_bounds,
                             unsigned &c_hit, unsigned &c_p, unsigned &c_s) {
    PixelOut out;
    V3 ro, rd;
    float depth;
    size_t gidx;
    rm_setup_ray(U, x, yl, pose, ro, rd, depth, gidx, out.oidx);
    const float sf = rm_vgpr(U.step_factor), kc = rm_vgpr(.0025f);

    // ---- raymarch (cginc:367-392) ----
    bool active = in_bounds && U.n_steps > 0;
    float dm = 0.0f, adts = __builtin_inff(), contact = 0.0f;
    unsigned evals = 0;
#ifdef RM_SKIP_STATS
    sc.st_phase = 1;
#endif
    for (int i = 0; i < U.n_steps; i++) {
        if (!__any(active)) break;
        const V3 cp = mk3(ro.x + rd.x * dm, ro.y + rd.y * dm, ro.z + rd.z * dm);
        sc.prepare(active, cp);
        const float v = sc.eval(cp);
        if (active) {
            adts = fabsf(v);
            evals++;
            dm = dm + adts * sf;
            contact = dm * kc;
            if (adts < contact || dm > depth) active = false;
        }
    }
    const bool hit = in_bounds && adts < contact;
    c_p += evals;
    c_hit += hit ? 1u : 0u;
    out.depth = dm;
    out.hit = hit ? 1u : 0u;
    out.rgba = 0;
    out.steps = 0;

    if (U.mode == RM_MODE_FRAG) {
        const V3 L = mk3(rm_vgpr(U.light[0]), rm_vgpr(U.light[1]), rm_vgpr(U.light[2]));
        // ---- normal: six taps within _NormalSampleScale of the hit point, one set of bounds for all of them
        // (shader:63-67); colour: the whole table (shader:193) ----
        const V3 hp = mk3(ro.x + rd.x * dm, ro.y + rd.y * dm, ro.z + rd.z * dm);
        V3 n = mk3(0.0f, 0.0f, 0.0f), raw = n;
        V4 s;
        s.x = s.y = s.z = s.w = 0.0f;
        if (__any(hit)) {
            sc.prepare(hit, hp, U.nss);
            const V3 taps = rm_normal_raw(U, Taps<SKIP>{sc}, hp);
            if (hit) {
                raw = taps;
                n = rm_normalize(raw);
                s = strict.full(hp);
            }
            sc.n_visited += (unsigned)sc.coop.part.n; sc.n_possible += (unsigned)sc.coop.part.n;      // the colour evaluation walks the whole table
        }
        // ---- getshadow (shader:102-123), hit lanes only ----
        const float c0 = contact;
        const V3 p = mk3(hp.x + (n.x * c0) * 2.0f, hp.y + (n.y * c0) * 2.0f, hp.z + (n.z * c0) * 2.0f);
        bool sactive = hit && U.n_steps > 0;
        float sdm = 0.0f, res = 1.0f;
        unsigned sev = 0;
#ifdef RM_SKIP_STATS
        sc.st_phase = 2;
#endif
        for (int i = 0; i < U.n_steps; i++) {
            if (!__any(sactive)) break;
            const V3 cp = mk3(p.x + L.x * sdm, p.y + L.y * sdm, p.z + L.z * sdm);
            sc.prepare(sactive, cp);
            const float dts = sc.eval(cp);
            if (sactive) {
                sev++;
                res = fminf(res, 16.0f * dts / sdm);
                const float a = fabsf(dts);
                sdm = sdm + a * sf;
                const float ct = sdm * kc;
                if (a < ct || sdm > U.shadow_depth) sactive = false;
            }
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

}  // namespace rmx
