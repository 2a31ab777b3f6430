// rm_k_skip.hip -- tables on the shared-tile kernel with exact culling (rm_skip.h): one launch block per
// 8x8 pixel tile, its four waves march the same 64 rays and each walks a quarter of the table, skipping the left subtrees
// that provably cannot change a bit of the result.  Frames are bit-identical to the other exact kernels'.  Compiled
// without the SLP vectoriser like the other table kernels.
#include "rm_kernels_common.h"
#include "rm_coop.h"
#include "rm_skip.h"

namespace {

#ifndef RM_SKIP_WAVES
#define RM_SKIP_WAVES 5
#endif
// (with op bits a fifth resident block would cost over 100 bytes of scratch, as in rm_k_coop.hip: those stay at four waves)
template <bool ROT, int SCL, bool BIG, bool CSG>
__global__ __launch_bounds__(256, CSG ? 4 : RM_SKIP_WAVES) void rm_coop_kernelN_skip(const RmParams U) {
    __shared__ float s_xw[2 * 4 * 64];
    __shared__ float s_x4[4 * 4 * 64];
    PrimTable<true> t;
    t.T = rm_stage_table(U, 0);                                           // ends with a __syncthreads()
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const SceneNCoop<ROT, SCL> coop = rm_make_coop_scene<ROT, SCL>(U, t, (RM_LDS float *)s_xw, (RM_LDS float *)s_x4, CSG);
    const int chunk = rm_coop_chunk(U.n_prims);
    rmx::SceneNSkip<ROT, SCL, BIG, CSG> sc(coop, chunk);
#ifdef RM_SKIP_STATS
    __shared__ unsigned s_stat[8];
    sc.st_lds = (RM_LDS unsigned *)s_stat;
#endif
    unsigned c_hit = 0, c_p = 0, c_s = 0;
    const int x = blockIdx.x * 8 + (int)(lane & 7u), yl = blockIdx.y * 8 + (int)(lane >> 3);
    const bool in_bounds = x < U.w && yl < U.rows;
    PixelOut po = rmx::compute_pixel(U, sc, coop, in_bounds ? x : U.w - 1, in_bounds ? yl : U.rows - 1, (int)blockIdx.z, in_bounds,
                                     c_hit, c_p, c_s);
    if (U.leaf_stats && lane == 0) {
        const unsigned bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        U.leaf_stats[(bid * 4u + wv) * 2u] = sc.n_visited;
        U.leaf_stats[(bid * 4u + wv) * 2u + 1u] = sc.n_possible;
    }
    if (wv == 0) {
        if (in_bounds) rm_store_pixel(U, po);                             // the four waves hold the same pixels
        c_hit = rm_wave_sum(c_hit); c_p = rm_wave_sum(c_p); c_s = rm_wave_sum(c_s);
#ifdef RM_SKIP_STATS           // the counters carry the statistics instead
        c_hit = sc.st_max; c_p = sc.st_sum; c_s = sc.st_n;
#endif
        const unsigned bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (lane < 3) U.partials[RM_SLOT_WORDS * bid + lane] = lane == 0 ? c_hit : lane == 1 ? c_p : c_s;
    }
}

template <bool ROT, int SCL>
hipError_t launch_skip_variant(const RmParams &U, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)U.n_prims * sizeof(RmPrimDev);
    const bool big = U.n_prims > 256, csg = (U.scene_flags & RM_SF_CSG) != 0;
    if (big && csg)  hipLaunchKernelGGL((rm_coop_kernelN_skip<ROT, SCL, true, true>), grid, dim3(256), lds, stream, U);
    else if (big)    hipLaunchKernelGGL((rm_coop_kernelN_skip<ROT, SCL, true, false>), grid, dim3(256), lds, stream, U);
    else if (csg)    hipLaunchKernelGGL((rm_coop_kernelN_skip<ROT, SCL, false, true>), grid, dim3(256), lds, stream, U);
    else             hipLaunchKernelGGL((rm_coop_kernelN_skip<ROT, SCL, false, false>), grid, dim3(256), lds, stream, U);
    return hipGetLastError();
}

}  // namespace

bool rm_skip_applies(const RmParams &U) {
    if (U.body != RM_BODY_PRIMS || U.lds_prims != U.n_prims || U.n_prims < 3) return false;
    const int chunk = rm_coop_chunk(U.n_prims);
    // (tables with op bits run at four waves per SIMD, where quarters of 16 leaves lose: 37 primitives 1.78 -> 1.96 ms)
    const int min_chunk = (U.scene_flags & RM_SF_CSG) ? 2 * RM_SKIP_MIN_CHUNK : RM_SKIP_MIN_CHUNK;
    return chunk >= min_chunk && chunk <= RM_SKIP_MAX_CHUNK;
}

// tables staged whole in LDS whose quarter is 16 to 128 primitives (33 to 512 primitives; from 65 with op bits); same grid and
// counter slots as rm_launch_coop_tiles
hipError_t rm_launch_coop_skip_tiles(const RmParams &U, hipStream_t stream) {
    if (U.w <= 0 || U.rows <= 0 || U.n_poses <= 0 || !rm_skip_applies(U)) return hipErrorInvalidValue;
    dim3 grid((unsigned)((U.w + 7) / 8), (unsigned)((U.rows + 7) / 8), (unsigned)U.n_poses);
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
    RM_DISPATCH_VARIANT(launch_skip_variant, U, U, grid, stream);
}
