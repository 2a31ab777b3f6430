// rm_k_cull.hip -- the opt-in tolerance kernel for table scenes: per-step culling against bounding spheres (rm_cull.h).
// Never selected automatically.  Compiled without the SLP vectoriser like the other table kernels (the operands come from
// LDS records in a fixed register order, so packed pairs cost register moves).
#include "rm_kernels_common.h"
#include "rm_fast.h"
#include "rm_cull.h"

namespace {

// N-primitive smooth-union scenes: one launch block per 16x16 pixels, every wave culls for its own 8x8 tile
template <bool ROT, bool SCALED, bool FASTM>
__global__ __launch_bounds__(RM_BLOCK) void rm_tile_kernelN_cull(const RmParams U) {
    __shared__ unsigned int s_cnt[3];
    const int tid = threadIdx.x;
    if (tid < 3) s_cnt[tid] = 0;
    PrimTable<true> t;
    t.T = rm_stage_table(U, 0, (unsigned)U.cull_extra16);               // ends with a __syncthreads()
    rmc::SceneNCull<ROT, SCALED, FASTM> sc;
    sc.tiny = rm_vgpr(RM_SQRT_TINY);
    sc.tab = t; sc.n = U.n_prims; sc.n_words = (U.n_prims + 63) >> 6;
    extern __shared__ __attribute__((aligned(16))) unsigned s_dyn[];
    sc.cr = (const RM_LDS rmc::rm_f4v *)(s_dyn + (size_t)U.n_prims * (sizeof(RmPrimDev) / 4));
    sc.ir = (const RM_LDS float *)(s_dyn + (size_t)U.n_prims * (sizeof(RmPrimDev) / 4 + 4));
    sc.slack = U.cull_slack;
    SceneN<ROT, SCALED ? RM_SCL_RCP : RM_SCL_UNIT, true> strict(t, U.n_prims, false, 0u);
    unsigned c_hit = 0, c_p = 0, c_s = 0;
    const int lane = tid & 63, wv = tid >> 6;
    const int x  = blockIdx.x * (8 * RM_BWX) + (wv % RM_BWX) * 8 + (lane & 7);
    const int yl = blockIdx.y * (8 * RM_BWY) + (wv / RM_BWX) * 8 + (lane >> 3);
    const bool in_bounds = x < U.w && yl < U.rows;
    PixelOut po = rmc::compute_pixel(U, sc, strict, in_bounds ? x : U.w - 1, in_bounds ? yl : U.rows - 1, (int)blockIdx.z, in_bounds,
                                     c_hit, c_p, c_s);
    if (in_bounds) rm_store_pixel(U, po);
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
    }
}

template <bool ROT, bool SCALED>
hipError_t launch_cull_variant(const RmParams &U, dim3 grid, hipStream_t stream) {
    const size_t lds = (size_t)U.n_prims * sizeof(RmPrimDev) + (size_t)U.cull_extra16 * 16;
    if (U.cull_fast_math) hipLaunchKernelGGL((rm_tile_kernelN_cull<ROT, SCALED, true>), grid, dim3(RM_BLOCK), lds, stream, U);
    else                  hipLaunchKernelGGL((rm_tile_kernelN_cull<ROT, SCALED, false>), grid, dim3(RM_BLOCK), lds, stream, U);
    return hipGetLastError();
}

}  // namespace

// N-primitive scenes without op bits whose table (and its bounding spheres) is staged in LDS: U.lds_prims == U.n_prims,
// U.cull_extra16 > 0
hipError_t rm_launch_cull_tiles(const RmParams &U, hipStream_t stream) {
    if (U.w <= 0 || U.rows <= 0 || U.n_poses <= 0 || U.n_prims <= 0 || U.n_prims > 64 * RM_CULL_WORDS || U.lds_prims != U.n_prims) return hipErrorInvalidValue;
    dim3 grid((unsigned)((U.w + 8 * RM_BWX - 1) / (8 * RM_BWX)), (unsigned)((U.rows + 8 * RM_BWY - 1) / (8 * RM_BWY)), (unsigned)U.n_poses);
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
    const bool rot = !(U.scene_flags & RM_SF_IDENT_ROT), unit = (U.scene_flags & RM_SF_UNIT_SCALE) != 0;
    if (rot) return unit ? launch_cull_variant<true, false>(U, grid, stream) : launch_cull_variant<true, true>(U, grid, stream);
    return unit ? launch_cull_variant<false, false>(U, grid, stream) : launch_cull_variant<false, true>(U, grid, stream);
}
