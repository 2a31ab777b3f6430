#pragma once
// rm_coop.h -- a table scene evaluated by the four waves of a workgroup together: every wave holds the SAME 64 sample
// points (one per lane) and evaluates a quarter of the primitive table.
//
// The scene value is the balanced pairwise smin tree over the table in index order (cginc:363 lifted to N leaves).  A chunk
// of C = 2^k consecutive leaves that starts at a multiple of C is a subtree of that tree, so with C = pow2ceil(n) / 4 the
// table splits into 3 or 4 chunks whose values S0..S3 meet in the top two levels: smin(smin(S0, S1), smin(S2, S3)), or
// smin(smin(S0, S1), S2) when the last chunk is empty (an odd element is carried up unchanged).  Every wave evaluates its
// chunk with the same code as the one-wave kernels, the four values go through LDS, and every wave finishes the tree
// itself, so all four hold bit-identical states and take every branch together.  Same operations in the same order as a
// strict binary32 evaluation of the shader source.  Nodes may be smax (RM_OP_SMAX): the op of a node is carried by the
// first leaf of its right operand, which for the three top nodes are leaves C, 3C and 2C.
#include "rm_kernels_common.h"

// leaves per wave when the four waves of a workgroup split a table of n >= 3 primitives: pow2ceil(n) / 4
__host__ __device__ inline int rm_coop_chunk(int n_prims) { return (1 << (32 - __builtin_clz((unsigned)n_prims - 1u))) >> 2; }

namespace {

template <bool ROT, int SCL>
struct SceneNCoop {
    typedef SceneN<ROT, SCL, true> Part;
    Part part;                            // this wave's chunk as a tree of its own
    int n_chunks;                         // 3 or 4
    unsigned wv, lane;
    RM_LDS float *xw;                     // [2][4][64]: distances, double-buffered so that one barrier per evaluation is enough
    RM_LDS float *x4;                     // [4][4][64]: rgb + distance of the colour evaluation (one per pixel)
    // ops of the three top nodes (+1 smin, -1 smax): a node's op is that of the first leaf of its right operand, i.e. of
    // leaf C for (S0, S1), 3C for (S2, S3) and 2C for the root
    float o01, o23, otop;
    mutable unsigned parity;

    template <class T> RM_DI T top(T s0, T s1, T s2, T s3) const {
        if (part.csg) {
            const T s01 = Part::template comb<true>(s0, s1, o01);
            if (n_chunks == 4) return Part::template comb<true>(s01, Part::template comb<true>(s2, s3, o23), otop);
            return Part::template comb<true>(s01, s2, otop);
        }
        const T s01 = Part::template comb<false>(s0, s1, 1.0f);
        if (n_chunks == 4) return Part::template comb<false>(s01, Part::template comb<false>(s2, s3, 1.0f), 1.0f);
        return Part::template comb<false>(s01, s2, 1.0f);
    }
    RM_DI float w(V3 p) const {
        float mine = 0.0f;
        if (part.n > 0) mine = part.w(p);
        return finish(mine);
    }
    // the value of this wave's chunk meets the other three
    RM_DI float finish(float mine) const {
        RM_LDS float *b = xw + parity * 256u;
        b[wv * 64u + lane] = mine;
        __syncthreads();
        const float r = top<float>(b[lane], b[64u + lane], b[128u + lane], b[192u + lane]);
        // the buffer written next is the other one: a wave that runs ahead cannot overwrite what a slower one still reads,
        // because it meets that wave at the next barrier before it returns to this buffer
        parity ^= 1u;
        return r;
    }
    RM_DI float wm(V3 p) const { return w(p); }
    RM_DI V4 full(V3 p) const {
        V4 m; m.x = m.y = m.z = m.w = 0.0f;
        if (part.n > 0) m = part.full(p);
        RM_LDS float *b = x4 + wv * 256u;
        b[lane] = m.x; b[64u + lane] = m.y; b[128u + lane] = m.z; b[192u + lane] = m.w;
        __syncthreads();
        V4 s[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            RM_LDS const float *q = x4 + k * 256;
            s[k].x = q[lane]; s[k].y = q[64u + lane]; s[k].z = q[128u + lane]; s[k].w = q[192u + lane];
        }
        return top<V4>(s[0], s[1], s[2], s[3]);
    }
};

// the scene object of the calling wave: xw = 2 * 4 * 64 floats, x4 = 4 * 4 * 64 floats of this workgroup's LDS
// (csg: the table carries op bits; a compile-time constant at the call keeps the other tree out of the kernel)
template <bool ROT, int SCL>
RM_DI SceneNCoop<ROT, SCL> rm_make_coop_scene(const RmParams &U, PrimTable<true> t, RM_LDS float *xw, RM_LDS float *x4, bool csg) {
    // the wave index is the same in every lane: as a scalar, the chunk bounds and with them the whole table walk stay on
    // the scalar unit (left as a vector value, the walk becomes a divergent loop with exec-mask bookkeeping)
    const unsigned lane = threadIdx.x & 63u, wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunk = rm_coop_chunk(U.n_prims);               // >= 1 for n >= 3
    const int lo = (int)wv * chunk;
    int cnt = U.n_prims - lo;
    cnt = cnt < 0 ? 0 : (cnt > chunk ? chunk : cnt);
    PrimTable<true> tc;
    tc.T = t.T + lo * (int)(sizeof(RmPrimDev) / 16);
    // a chunk's own tree takes its node ops from the records it walks and, for a partial last chunk, from the low bits of
    // U.tail_ops: below the chunk level the merges of the whole-table walk and of the chunk's walk are the same ones
    const int n_chunks = (U.n_prims + chunk - 1) / chunk;
    float o01 = 1.0f, o23 = 1.0f, otop = 1.0f;
    if (csg) {
        o01 = t.get(chunk).osgn;
        otop = t.get(2 * chunk).osgn;
        if (n_chunks == 4) o23 = t.get(3 * chunk).osgn;
    }
    return SceneNCoop<ROT, SCL>{SceneN<ROT, SCL, true>(tc, cnt, csg, U.tail_ops), n_chunks, wv, lane, xw, x4, o01, o23, otop, 0u};
}

}  // namespace
