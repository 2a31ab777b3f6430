// rm_k_coop.hip -- table scenes, one launch block per 8x8 pixel tile: the four waves of a block march the SAME 64 rays and
// each evaluates a quarter of the primitive table.
//
// The scene value is the balanced pairwise smin tree over the table in index order (cginc:363 lifted to N leaves).  A chunk
// of C = 2^k consecutive leaves that starts at a multiple of C is a subtree of that tree, so with C = pow2ceil(n) / 4 the
// table splits into 3 or 4 chunks whose values S0..S3 meet in the top two levels: smin(smin(S0, S1), smin(S2, S3)), or
// smin(smin(S0, S1), S2) when the last chunk is empty (an odd element is carried up unchanged).  Every wave evaluates its
// chunk with the same code as the one-wave kernels, the four values go through LDS, and every wave finishes the tree
// itself, so all four hold bit-identical march states and take every branch together.  Same operations in the same order
// as a strict binary32 evaluation of the shader source: the frames are bit-identical to the other exact kernels'.  Nodes may be smax (RM_OP_SMAX): the op of a
// node is carried by the first leaf of its right operand, which for the three top nodes are leaves C, 3C and 2C.
//
// Why: a 256-primitive evaluation is ~18 000 VALU instructions; with one wave per tile the longest tile of a frame is a
// chain of up to 400 of them and the four waves of a 16x16 block finish far apart, which leaves wave slots empty (2.6 of 4
// resident on average).  Here a block's waves finish together and the chain is four times shorter; the price is one
// LDS exchange and one barrier per evaluation (a few dozen cycles against ~18 000).
#include "rm_kernels_common.h"

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

// CSG: the table carries op bits.  Kept apart so that the plain-union kernel, which is the one the big tables of the
// benchmark use, does not carry the smax paths in its register budget (4 waves per SIMD; 3 with op bits).
template <bool ROT, int SCL, bool CSG>
__global__ __launch_bounds__(256, CSG ? 3 : 4) void rm_coop_kernelN(const RmParams U) {
    __shared__ float s_xw[2 * 4 * 64];
    __shared__ float s_x4[4 * 4 * 64];
    PrimTable<true> t;
    t.T = rm_stage_table(U, 0);                                           // ends with a __syncthreads()
    const int tid = threadIdx.x;
    const unsigned lane = tid & 63, wv = tid >> 6;
    // chunk = pow2ceil(n) / 4, n >= 8
    const int chunk = (1 << (32 - __builtin_clz((unsigned)U.n_prims - 1u))) >> 2;
    const int lo = (int)wv * chunk;
    int cnt = U.n_prims - lo;
    cnt = cnt < 0 ? 0 : (cnt > chunk ? chunk : cnt);
    PrimTable<true> tc;
    tc.T = t.T + lo * (int)(sizeof(RmPrimDev) / 16);
    // a chunk's own tree takes its node ops from the records it walks and, for a partial last chunk, from the low bits of
    // U.tail_ops: below the chunk level the merges of the whole-table walk and of the chunk's walk are the same ones
    constexpr bool csg = CSG;
    const int n_chunks = (U.n_prims + chunk - 1) / chunk;
    float o01 = 1.0f, o23 = 1.0f, otop = 1.0f;
    if (csg) {
        o01 = t.get(chunk).osgn;
        otop = t.get(2 * chunk).osgn;
        if (n_chunks == 4) o23 = t.get(3 * chunk).osgn;
    }
    SceneNCoop<ROT, SCL> sc{SceneN<ROT, SCL, true>(tc, cnt, csg, U.tail_ops), n_chunks, wv, lane,
                            (RM_LDS float *)s_xw, (RM_LDS float *)s_x4, o01, o23, otop, 0u};
    unsigned c_hit = 0, c_p = 0, c_s = 0;
    const int x = blockIdx.x * 8 + (int)(lane & 7u), yl = blockIdx.y * 8 + (int)(lane >> 3);
    if (x < U.w && yl < U.rows) {
        PixelOut po = rm_compute_pixel(U, sc, x, yl, (int)blockIdx.z, c_hit, c_p, c_s);
        if (wv == 0) rm_store_pixel(U, po);                               // the four waves hold the same pixels
    }
    if (wv == 0) {
        c_hit = rm_wave_sum(c_hit); c_p = rm_wave_sum(c_p); c_s = rm_wave_sum(c_s);
        const unsigned bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (lane < 3) U.partials[RM_SLOT_WORDS * bid + lane] = lane == 0 ? c_hit : lane == 1 ? c_p : c_s;
    }
}

template <bool ROT, int SCL>
hipError_t launch_coop_variant(const RmParams &U, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)U.n_prims * sizeof(RmPrimDev);
    if (U.scene_flags & RM_SF_CSG) hipLaunchKernelGGL((rm_coop_kernelN<ROT, SCL, true>), grid, dim3(256), lds, stream, U);
    else                           hipLaunchKernelGGL((rm_coop_kernelN<ROT, SCL, false>), grid, dim3(256), lds, stream, U);
    return hipGetLastError();
}

}  // namespace

unsigned rm_coop_blocks(int w, int rows, int n_poses) {
    return (unsigned)((w + 7) / 8) * (unsigned)((rows + 7) / 8) * (unsigned)n_poses;
}

// table scenes staged whole in LDS (U.lds_prims == U.n_prims), RM_COOP_MIN_PRIMS <= n
hipError_t rm_launch_coop_tiles(const RmParams &U, hipStream_t stream) {
    if (U.w <= 0 || U.rows <= 0 || U.n_poses <= 0 || U.n_prims < RM_COOP_MIN_PRIMS || U.lds_prims != U.n_prims ||
        U.body != RM_BODY_PRIMS) return hipErrorInvalidValue;
    dim3 grid((unsigned)((U.w + 7) / 8), (unsigned)((U.rows + 7) / 8), (unsigned)U.n_poses);
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
    RM_DISPATCH_VARIANT(launch_coop_variant, U, U, grid, stream);
}
