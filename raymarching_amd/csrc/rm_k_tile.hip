// rm_k_tile.hip -- one launch block per image tile: the reference scene, table scenes;
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

// clears n 32-bit words: the claim counters of a launch that must not rely on an earlier launch having cleared them.  A
// kernel rather than a memset so that, recorded into a graph, it is a kernel node ordered like any other launch.
__global__ __launch_bounds__(256) void rm_zero_words_kernel(uint32_t *__restrict__ p, unsigned n) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// RGBA8 -> RGB8: the alpha the blit writes is the constant 255 (fixed4(col, 1), shader:214), so a frame
// that travels over xGMI to the displaying rank carries 3 bytes per pixel.  One thread = 4 pixels:
// 16 bytes in, 12 bytes out, both as dwords.
__global__ __launch_bounds__(256) void rm_pack_rgb_kernel(const uint32_t *__restrict__ rgba, uint32_t *__restrict__ rgb, size_t n_quads) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_quads) return;
    const uint4 p = reinterpret_cast<const uint4 *>(rgba)[q];
    const uint32_t a = p.x & 0xffffffu, b = p.y & 0xffffffu, c = p.z & 0xffffffu, d = p.w & 0xffffffu;
    uint32_t *o = rgb + q * 3;
    o[0] = a | (b << 24);
    o[1] = (b >> 8) | (c << 16);
    o[2] = (c >> 16) | (d << 8);
}
__global__ __launch_bounds__(256) void rm_unpack_rgb_kernel(const uint32_t *__restrict__ rgb, uint32_t *__restrict__ rgba, size_t n_quads) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_quads) return;
    const uint32_t *i = rgb + q * 3;
    const uint32_t x = i[0], y = i[1], z = i[2];
    uint4 p;
    p.x = (x & 0xffffffu) | 0xff000000u;
    p.y = ((x >> 24) | ((y & 0xffffu) << 8)) | 0xff000000u;
    p.z = ((y >> 16) | ((z & 0xffu) << 16)) | 0xff000000u;
    p.w = (z >> 8) | 0xff000000u;
    reinterpret_cast<uint4 *>(rgba)[q] = p;
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

hipError_t rm_zero_words(void *p, unsigned n_words, hipStream_t stream) {
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(rm_zero_words_kernel, dim3((n_words + 255) / 256), dim3(256), 0, stream, (uint32_t *)p, n_words);
    return hipGetLastError();
}

hipError_t rm_pack_rgb(const void *rgba, void *rgb, size_t n_pixels, bool unpack, hipStream_t stream) {
    const size_t n_quads = n_pixels / 4;
    if (n_quads == 0) return hipSuccess;
    const size_t blocks = (n_quads + 255) / 256;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (unpack) hipLaunchKernelGGL(rm_unpack_rgb_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint32_t *)rgba, (uint32_t *)rgb, n_quads);
    else        hipLaunchKernelGGL(rm_pack_rgb_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint32_t *)rgba, (uint32_t *)rgb, n_quads);
    return hipGetLastError();
}

int rm_variant_index(const RmParams &U) { bool r; int s; return U.body != RM_BODY_PRIMS ? 6 : rm_scene_variant(U, r, s); }

hipError_t rm_launch_tiles(const RmParams &U, hipStream_t stream) {
    if (U.w <= 0 || U.rows <= 0 || U.n_poses <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((U.w + 8 * RM_BWX - 1) / (8 * RM_BWX)), (unsigned)((U.rows + 8 * RM_BWY - 1) / (8 * RM_BWY)), (unsigned)U.n_poses);
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
    if (U.fx_flags != 0 || U.fx_view != 0 || U.light_model != 0 || U.dom_op != RM_DOM_NONE || (U.mode == RM_MODE_FRAG && U.shadow_mode != 0)) return rm_launch_tiles_fx(U, grid.x, grid.y, grid.z, stream);
    if (U.body != RM_BODY_PRIMS) return rm_launch_tiles_body(U, grid.x, grid.y, grid.z, stream);
    RM_DISPATCH_VARIANT(launch_variant, U, U, grid, stream);
}
