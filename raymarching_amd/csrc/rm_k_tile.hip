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

// ---- frame transport: only what the receiver cannot reconstruct ----------------------------------------
// A finished frame is the background (the constant _BackgroundColor or the source image, both of which the
// displaying rank holds) everywhere but in the 8x8 tiles that contain a hit pixel.  The compactor appends one
// record per such tile: the tile's index (frame * tiles_per_frame + tile), then its 64 pixels as RGB8
// (RM_RECORD_WORDS = 1 + 48 dwords).  The blit's alpha is the constant 255 and is not sent.
//
// One lane looks at one tile's 64 mask bytes; a block of 1024 lanes (1024 consecutive tiles) reserves its
// records with ONE atomic on the counter (a pose of 7680x4320 is 506 blocks), then every wave copies its hit
// tiles cooperatively, one lane per pixel, packing four pixels into three dwords across lanes.
#define RM_RECORD_WORDS 49
__global__ __launch_bounds__(1024) void rm_compact_tiles_kernel(const uint32_t *__restrict__ rgba, const uint8_t *__restrict__ hit,
                                                                int w, int h, unsigned tiles_x, unsigned tiles_y, unsigned n_tiles_total,
                                                                uint32_t *__restrict__ records, uint32_t *__restrict__ counter) {
    __shared__ unsigned s_wave_count[16];
    __shared__ unsigned s_block_base;
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned t = blockIdx.x * 1024u + threadIdx.x;
    const unsigned per_frame = tiles_x * tiles_y;
    bool any = false;
    if (t < n_tiles_total) {
        const unsigned frame = t / per_frame, t2 = t - frame * per_frame;
        const unsigned ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const uint8_t *m = hit + (size_t)frame * (size_t)w * (size_t)h;
        for (unsigned r = 0; r < 8; r++) {
            const unsigned y = ty * 8 + r;
            if (y >= (unsigned)h) break;
            const size_t row = (size_t)y * (size_t)w + (size_t)tx * 8;
            if (tx * 8 + 8 <= (unsigned)w && ((uintptr_t)(m + row) & 7u) == 0) {
                any = any || *reinterpret_cast<const unsigned long long *>(m + row) != 0ull;
            } else {
                for (unsigned c = 0; c < 8 && tx * 8 + c < (unsigned)w; c++) any = any || m[row + c] != 0;
            }
        }
    }
    const unsigned long long mask = __ballot(any);
    if (lane == 0) s_wave_count[wv] = (unsigned)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
        for (int i = 0; i < 16; i++) { const unsigned c = s_wave_count[i]; s_wave_count[i] = total; total += c; }
        s_block_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    unsigned slot = s_block_base + s_wave_count[wv];
    unsigned long long rest = mask;
    while (rest) {
        const unsigned src_lane = (unsigned)__builtin_ctzll(rest);
        rest &= rest - 1;
        const unsigned tile = blockIdx.x * 1024u + wv * 64u + src_lane;
        const unsigned frame = tile / per_frame, t2 = tile - frame * per_frame;
        const unsigned ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const unsigned x = tx * 8 + (lane & 7u), y = ty * 8 + (lane >> 3);
        uint32_t px = 0;
        if (x < (unsigned)w && y < (unsigned)h) px = rgba[((size_t)frame * (size_t)h + (size_t)y) * (size_t)w + (size_t)x] & 0xffffffu;
        // lanes 0..47 each build one dword of the 192-byte RGB stream: group g = four pixels -> three dwords
        const unsigned g = lane / 3u, r = lane - g * 3u;
        const uint32_t lo = (uint32_t)__shfl((int)px, (int)((4u * g + r) & 63u)), hi = (uint32_t)__shfl((int)px, (int)((4u * g + r + 1u) & 63u));
        uint32_t *rec = records + (size_t)slot * RM_RECORD_WORDS;
        if (lane < 48) rec[1 + lane] = r == 0 ? (lo | (hi << 24)) : r == 1 ? ((lo >> 8) | (hi << 16)) : ((lo >> 16) | (hi << 8));
        if (lane == 48) rec[0] = tile;
        slot++;
    }
}

// inverse: one wave per record writes the tile's 64 pixels (alpha 255) over a frame that already holds the background
__global__ __launch_bounds__(256) void rm_expand_tiles_kernel(const uint32_t *__restrict__ records, unsigned n_records, int w, int h,
                                                              unsigned tiles_x, unsigned tiles_y, unsigned n_frames, uint32_t *__restrict__ rgba) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned rec_i = (blockIdx.x * 256u + threadIdx.x) >> 6;
    if (rec_i >= n_records) return;
    const uint32_t *rec = records + (size_t)rec_i * RM_RECORD_WORDS;
    const unsigned tile = rec[0];
    const unsigned per_frame = tiles_x * tiles_y;
    const unsigned frame = tile / per_frame, t2 = tile - frame * per_frame;
    if (frame >= n_frames) return;                       // a record that does not belong to this batch: ignored
    const unsigned ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
    // pixel p = bytes 3p..3p+2 of the stream: dwords (3p)>>2 and the next
    const unsigned b = 3u * lane, d0 = b >> 2, sh = (b & 3u) * 8u;
    const uint32_t a = rec[1 + d0], c = d0 + 1 < 48 ? rec[2 + d0] : 0u;
    const uint32_t px = (sh ? ((a >> sh) | (c << (32u - sh))) : a) & 0xffffffu;
    const unsigned x = tx * 8 + (lane & 7u), y = ty * 8 + (lane >> 3);
    if (x < (unsigned)w && y < (unsigned)h) rgba[((size_t)frame * (size_t)h + (size_t)y) * (size_t)w + (size_t)x] = px | 0xff000000u;
}

// every frame of the batch starts as the source image with the alpha the blit writes
__global__ __launch_bounds__(256) void rm_fill_from_src_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t px, unsigned n_frames) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= px) return;
    const uint32_t v = src[i] | 0xff000000u;
    for (unsigned f = 0; f < n_frames; f++) dst[(size_t)f * px + i] = v;
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

hipError_t rm_pack_rgb(const void *rgba, void *rgb, size_t n_pixels, bool unpack, hipStream_t stream) {
    const size_t n_quads = n_pixels / 4;
    if (n_quads == 0) return hipSuccess;
    const size_t blocks = (n_quads + 255) / 256;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    if (unpack) hipLaunchKernelGGL(rm_unpack_rgb_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint32_t *)rgba, (uint32_t *)rgb, n_quads);
    else        hipLaunchKernelGGL(rm_pack_rgb_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint32_t *)rgba, (uint32_t *)rgb, n_quads);
    return hipGetLastError();
}

unsigned rm_record_words() { return RM_RECORD_WORDS; }

hipError_t rm_compact_tiles(const void *rgba, const void *hit, int w, int h, unsigned n_frames, void *records, void *counter, hipStream_t stream) {
    const unsigned tx = (unsigned)(w + 7) / 8, ty = (unsigned)(h + 7) / 8;
    const unsigned long long total = (unsigned long long)tx * ty * n_frames;
    if (total == 0 || total > 0x7fffffffull) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(counter, 0, 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rm_compact_tiles_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(1024), 0, stream,
                       (const uint32_t *)rgba, (const uint8_t *)hit, w, h, tx, ty, (unsigned)total, (uint32_t *)records, (uint32_t *)counter);
    return hipGetLastError();
}

hipError_t rm_expand_tiles(const void *records, unsigned n_records, int w, int h, unsigned n_frames, void *rgba, hipStream_t stream) {
    if (n_records == 0) return hipSuccess;
    const unsigned tx = (unsigned)(w + 7) / 8, ty = (unsigned)(h + 7) / 8;
    hipLaunchKernelGGL(rm_expand_tiles_kernel, dim3((n_records + 3) / 4), dim3(256), 0, stream,
                       (const uint32_t *)records, n_records, w, h, tx, ty, n_frames, (uint32_t *)rgba);
    return hipGetLastError();
}

hipError_t rm_fill_from_src(const void *src, void *dst, size_t px, unsigned n_frames, hipStream_t stream) {
    const size_t blocks = (px + 255) / 256;
    if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rm_fill_from_src_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint32_t *)src, (uint32_t *)dst, px, n_frames);
    return hipGetLastError();
}

int rm_variant_index(const RmParams &U) { bool r; int s; return U.body != RM_BODY_PRIMS ? 6 : rm_scene_variant(U, r, s); }

hipError_t rm_launch_tiles(const RmParams &U, hipStream_t stream) {
    if (U.w <= 0 || U.rows <= 0 || U.n_poses <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((U.w + 8 * RM_BWX - 1) / (8 * RM_BWX)), (unsigned)((U.rows + 8 * RM_BWY - 1) / (8 * RM_BWY)), (unsigned)U.n_poses);
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
    if (U.fx_flags != 0 || U.fx_view != 0 || U.light_model != 0 || U.dom_op != RM_DOM_NONE || (U.mode == RM_MODE_FRAG && U.shadow_mode != 0)) return rm_launch_tiles_fx(U, grid.x, grid.y, grid.z, stream);
    if (U.body != RM_BODY_PRIMS) {
        hipLaunchKernelGGL(rm_tile_kernelF, grid, dim3(RM_BLOCK), 0, stream, U);
        return hipGetLastError();
    }
    RM_DISPATCH_VARIANT(launch_variant, U, U, grid, stream);
}
