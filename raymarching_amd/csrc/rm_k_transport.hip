// rm_k_transport.hip -- frame transport between GPUs: compaction of finished frames into tile records and its inverse.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rm_launch.h"

namespace {

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

// the tiles named by a set of records go back to the background (the source image where one is given, else bg8):
// what is left to undo in a frame that was expanded from those records before
__global__ __launch_bounds__(256) void rm_restore_tiles_kernel(const uint32_t *__restrict__ records, unsigned n_records, int w, int h,
                                                               unsigned tiles_x, unsigned tiles_y, unsigned n_frames,
                                                               const uint32_t *__restrict__ src, uint32_t bg8, uint32_t *__restrict__ rgba) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned rec_i = (blockIdx.x * 256u + threadIdx.x) >> 6;
    if (rec_i >= n_records) return;
    const unsigned tile = records[(size_t)rec_i * RM_RECORD_WORDS];
    const unsigned per_frame = tiles_x * tiles_y;
    const unsigned frame = tile / per_frame, t2 = tile - frame * per_frame;
    if (frame >= n_frames) return;
    const unsigned ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
    const unsigned x = tx * 8 + (lane & 7u), y = ty * 8 + (lane >> 3);
    if (x < (unsigned)w && y < (unsigned)h) {
        const size_t i = (size_t)y * (size_t)w + (size_t)x;
        rgba[(size_t)frame * (size_t)h * (size_t)w + i] = src ? (src[i] | 0xff000000u) : bg8;
    }
}

// every frame of the batch starts as the source image with the alpha the blit writes
__global__ __launch_bounds__(256) void rm_fill_from_src_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t px, unsigned n_frames) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= px) return;
    const uint32_t v = src[i] | 0xff000000u;
    for (unsigned f = 0; f < n_frames; f++) dst[(size_t)f * px + i] = v;
}

}  // namespace

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

hipError_t rm_restore_tiles(const void *records, unsigned n_records, int w, int h, unsigned n_frames, const void *src, uint32_t bg8,
                            void *rgba, hipStream_t stream) {
    if (n_records == 0) return hipSuccess;
    const unsigned tx = (unsigned)(w + 7) / 8, ty = (unsigned)(h + 7) / 8;
    hipLaunchKernelGGL(rm_restore_tiles_kernel, dim3((n_records + 3) / 4), dim3(256), 0, stream,
                       (const uint32_t *)records, n_records, w, h, tx, ty, n_frames, (const uint32_t *)src, bg8, (uint32_t *)rgba);
    return hipGetLastError();
}

hipError_t rm_fill_from_src(const void *src, void *dst, size_t px, unsigned n_frames, hipStream_t stream) {
    const size_t blocks = (px + 255) / 256;
    if (blocks == 0 || blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rm_fill_from_src_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint32_t *)src, (uint32_t *)dst, px, n_frames);
    return hipGetLastError();
}

