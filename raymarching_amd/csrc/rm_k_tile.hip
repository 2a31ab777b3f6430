// rm_k_tile.hip -- one launch block per image tile: the reference scene, table scenes, fractal bodies;
// counter reduction.
#include "rm_kernels_common.h"

namespace {

// sums the per-block counter slots of the last launch: out[0..2] = n_hit, primary evals, shadow evals
__global__ __launch_bounds__(1024) void rm_reduce_stats_kernel(const unsigned *__restrict__ partials, unsigned n_blocks,
                                                               unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s[3];
    if (threadIdx.x < 3) s[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long a0 = 0, a1 = 0, a2 = 0;
    for (unsigned b = threadIdx.x; b < n_blocks; b += 1024) {
        a0 += partials[RM_SLOT_WORDS * b]; a1 += partials[RM_SLOT_WORDS * b + 1]; a2 += partials[RM_SLOT_WORDS * b + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s[0], a0); atomicAdd(&s[1], a1); atomicAdd(&s[2], a2); }
    __syncthreads();
    if (threadIdx.x < 3) out[threadIdx.x] = s[threadIdx.x];
}

template <bool ROT, int SCL>
__global__ RM_LB void rm_tile_kernel4(const RmParams U) {
    Scene4<ROT, SCL> sc(U);
    rm_pixel(U, sc);
}

template <bool ROT, int SCL>
__global__ RM_LB void rm_tile_kernelN(const RmParams U) {
    if (U.lds_prims) {
        PrimTable<true> t; t.T = rm_stage_table(U, 0);
        SceneN<ROT, SCL, true> sc(t, U.n_prims, (U.scene_flags & RM_SF_CSG) != 0, U.tail_ops);
        rm_pixel(U, sc);
    } else {
        PrimTable<false> t; t.T = U.prim_table;
        SceneN<ROT, SCL, false> sc(t, U.n_prims, (U.scene_flags & RM_SF_CSG) != 0, U.tail_ops);
        rm_pixel(U, sc);
    }
}

// fractal scene() bodies: one kernel each, the slot-0 transform is specialised by wave-uniform branches
__global__ RM_LB void rm_tile_kernelF(const RmParams U) {
    SceneF sc(U);
    rm_pixel(U, sc);
}
template <bool ROT, int SCL>
hipError_t launch_variant(const RmParams &U, dim3 grid, hipStream_t stream) {
    if (U.n_prims > 0) hipLaunchKernelGGL((rm_tile_kernelN<ROT, SCL>), grid, dim3(RM_BLOCK), (size_t)U.lds_prims * sizeof(RmPrimDev), stream, U);
    else               hipLaunchKernelGGL((rm_tile_kernel4<ROT, SCL>), grid, dim3(RM_BLOCK), 0, stream, U);
    return hipGetLastError();
}

}  // namespace

unsigned rm_slot_words() { return RM_SLOT_WORDS; }

unsigned rm_tile_blocks(int w, int rows, int n_poses) {
    return (unsigned)((w + 8 * RM_BWX - 1) / (8 * RM_BWX)) * (unsigned)((rows + 8 * RM_BWY - 1) / (8 * RM_BWY)) * (unsigned)n_poses;
}

hipError_t rm_reduce_stats(const unsigned *partials, unsigned n_blocks, unsigned long long *out, hipStream_t stream) {
    hipLaunchKernelGGL(rm_reduce_stats_kernel, dim3(1), dim3(1024), 0, stream, partials, n_blocks, out);
    return hipGetLastError();
}

int rm_variant_index(const RmParams &U) { bool r; int s; return U.body != RM_BODY_PRIMS ? 6 : rm_scene_variant(U, r, s); }

hipError_t rm_launch_tiles(const RmParams &U, hipStream_t stream) {
    if (U.w <= 0 || U.rows <= 0 || U.n_poses <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((U.w + 8 * RM_BWX - 1) / (8 * RM_BWX)), (unsigned)((U.rows + 8 * RM_BWY - 1) / (8 * RM_BWY)), (unsigned)U.n_poses);
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
    if (U.fx_flags != 0 || U.fx_view != 0) return rm_launch_tiles_fx(U, grid.x, grid.y, grid.z, stream);
    if (U.body != RM_BODY_PRIMS) {
        hipLaunchKernelGGL(rm_tile_kernelF, grid, dim3(RM_BLOCK), 0, stream, U);
        return hipGetLastError();
    }
    RM_DISPATCH_VARIANT(launch_variant, U, U, grid, stream);
}
