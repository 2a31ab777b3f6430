// rm_k_coop.hip -- table scenes, one launch block per 8x8 pixel tile: the four waves of a block march the SAME 64 rays and
// each evaluates a quarter of the primitive table (rm_coop.h); the frames are bit-identical to the other exact kernels'.
//
// Why: a 256-primitive evaluation is ~18 000 VALU instructions; with one wave per tile the longest tile of a frame is a
// chain of up to 400 of them and the four waves of a 16x16 block finish far apart, which leaves wave slots empty (2.6 of 4
// resident on average).  Here a block's waves finish together and the chain is four times shorter; the price is one
// LDS exchange and one barrier per evaluation (a few dozen cycles against ~18 000).
#include "rm_kernels_common.h"
#include "rm_coop.h"

namespace {

// CSG: the table carries op bits.  Kept apart so that the plain-union kernel, which is the one the big tables of the
// benchmark use, does not carry the smax paths in its register budget (both fit 4 waves per SIMD).
// Waves per SIMD the register allocator is asked to fit.  A block's four waves meet at a barrier per evaluation, so a wave
// often waits for its slowest sibling; a fifth resident block covers that wait (stress scene 34.0 -> 31.1 ms, 128
// primitives 14.9 -> 13.8 ms) at the price of a few dwords of scratch outside the table walk.  With op bits the fifth block
// would cost over 100 bytes of scratch: those stay at four.
#ifndef RM_COOP_WAVES
#define RM_COOP_WAVES 5
#endif
template <bool ROT, int SCL, bool CSG>
__global__ __launch_bounds__(256, CSG ? 4 : RM_COOP_WAVES) void rm_coop_kernelN(const RmParams U) {
    __shared__ float s_xw[2 * 4 * 64];
    __shared__ float s_x4[4 * 4 * 64];
    PrimTable<true> t;
    t.T = rm_stage_table(U, 0);                                           // ends with a __syncthreads()
    const int tid = threadIdx.x;
    const unsigned lane = tid & 63, wv = tid >> 6;
    const SceneNCoop<ROT, SCL> sc = rm_make_coop_scene<ROT, SCL>(U, t, (RM_LDS float *)s_xw, (RM_LDS float *)s_x4, CSG);
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

// frag's optional lines, the light helpers, the shadow-map lookup, the domain operators and the step-count plane over a
// table (rm_compute_pixel_fx): rarely the bottleneck of anything, so one instantiation serves tables with and without op bits
template <bool ROT, int SCL>
__global__ __launch_bounds__(256) void rm_coop_kernelN_fx(const RmParams U) {
    __shared__ float s_xw[2 * 4 * 64];
    __shared__ float s_x4[4 * 4 * 64];
    PrimTable<true> t;
    t.T = rm_stage_table(U, 0);
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const SceneNCoop<ROT, SCL> sc = rm_make_coop_scene<ROT, SCL>(U, t, (RM_LDS float *)s_xw, (RM_LDS float *)s_x4,
                                                                   (U.scene_flags & RM_SF_CSG) != 0);
    unsigned c_hit = 0, c_p = 0, c_s = 0;
    const int x = blockIdx.x * 8 + (int)(lane & 7u), yl = blockIdx.y * 8 + (int)(lane >> 3);
    if (x < U.w && yl < U.rows) {
        PixelOut po = rm_compute_pixel_fx(U, SceneDom<SceneNCoop<ROT, SCL>>(sc, U), x, yl, (int)blockIdx.z, c_hit, c_p, c_s);
        po.steps = c_p | (c_s << 16);          // this thread's only pixel: the counters are the pixel's own
        if (wv == 0) { rm_store_pixel(U, po); rm_store_steps(U, po); }
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
    const bool fx = U.fx_flags != 0 || U.fx_view != 0 || U.light_model != 0 || U.dom_op != RM_DOM_NONE || U.dst_steps ||
                    (U.mode == RM_MODE_FRAG && U.shadow_mode != 0);
    if (fx) hipLaunchKernelGGL((rm_coop_kernelN_fx<ROT, SCL>), grid, dim3(256), lds, stream, U);
    else if (U.scene_flags & RM_SF_CSG) hipLaunchKernelGGL((rm_coop_kernelN<ROT, SCL, true>), grid, dim3(256), lds, stream, U);
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
