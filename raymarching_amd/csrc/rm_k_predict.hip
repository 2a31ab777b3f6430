// rm_k_predict.hip -- tile costs for a frame that has no previous frame to learn them from.
//
// The persistent kernel starts the tiles that were expensive in the context's previous frame first (rm_queue_body); the first
// frame of a context, a frame of another size and a frame after a camera cut have no such history, and their longest tile
// then starts wherever the scattered order puts it (a 1080p frame of the sample scene takes 141 us that way against 85 us).
// This kernel marches ONE ray per 8x8 tile -- through the tile's centre pixel -- until it has cost `threshold` scene
// evaluations (primary steps, 7 for a hit's normal and colour, shadow steps) or is finished, and lists the tiles whose ray
// reached the threshold in the heavy-tile set the frame kernel is about to read, exactly as a previous frame would have.
// On the sample scene a centre ray that costs 16 evaluations finds every tile whose most expensive ray costs 48 or more and
// nine in ten of those from 24 (the frame kernel's own threshold), and lists 5 % of the tiles (scripts/dev_predict_quality.py).
// The list only orders the queue: every tile is still rendered exactly once, so the image cannot depend on it.
#include "rm_kernels_common.h"

namespace {

// one thread per tile; a wave covers 64 consecutive tiles = two words of the tile bitmap, which it writes whole (no atomics);
// the list index costs one atomic per wave
template <bool ROT, int SCL>
__global__ __launch_bounds__(256) void rm_predict_kernel4(const RmParams U, unsigned threshold) {
    Scene4<ROT, SCL> sc(U);
    const RmQueue &Q = U.queue;
    const int lane = threadIdx.x & 63;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const bool valid = t < Q.n_tiles;
    unsigned cost = 0;
    if (valid) {
        const unsigned per_pose = Q.tiles_x * Q.tiles_y;
        const unsigned pose = t / per_pose, t2 = t - pose * per_pose;
        const int ty = (int)(t2 / Q.tiles_x), tx = (int)(t2 - (unsigned)ty * Q.tiles_x);
        const int x = tx * RM_TW + RM_TW / 2 < U.w ? tx * RM_TW + RM_TW / 2 : U.w - 1;
        const int yl = ty * RM_TH + RM_TH / 2 < U.rows ? ty * RM_TH + RM_TH / 2 : U.rows - 1;
        V3 ro, rd;
        float depth;
        size_t gidx, oidx;
        rm_setup_ray<0>(U, x, yl, (int)pose, ro, rd, depth, gidx, oidx);
        const float sf = rm_vgpr(U.step_factor);
        const int n = U.n_steps < (int)threshold ? U.n_steps : (int)threshold;
        const MarchResult r = rm_raymarch(sc, ro, rd, depth, n, sf);
        cost = (unsigned)r.evals;
        if (r.hit) {
            cost += 7u;                                    // six normal taps and the colour evaluation
            if (U.shadow_mode == 0 && cost < threshold) {
                const V3 L = mk3(rm_vgpr(U.light[0]), rm_vgpr(U.light[1]), rm_vgpr(U.light[2]));
                const V3 hp = mk3(ro.x + rd.x * r.length, ro.y + rd.y * r.length, ro.z + rd.z * r.length);
                const V3 nrm = rm_normalize(rm_normal_raw(U, sc, hp));
                const int left = (int)(threshold - cost);
                int sev = 0;
                (void)rm_getshadow(sc, hp, nrm, L, r.contact, U.shadow_depth, U.n_steps < left ? U.n_steps : left, sf, sev);
                cost += (unsigned)sev;
            }
        }
    }
    const bool listed = valid && cost >= threshold;
    const unsigned long long m = __ballot(listed);
    unsigned idx = 0;
    if (m) {
        if (lane == 0) idx = atomicAdd(&Q.heavy_count[Q.hv_cur], (unsigned)__builtin_popcountll(m));
        idx = __builtin_amdgcn_readfirstlane(idx) + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    }
    const bool keep = listed && idx < Q.heavy_cap;           // no room on the list: the tile stays in image order
    const unsigned long long mk = __ballot(keep);
    unsigned *bitmap = Q.bitmaps + (size_t)Q.hv_cur * Q.bitmap_words;
    const unsigned word = (t - (unsigned)lane) >> 5;         // the wave's first tile is a multiple of 64
    if (lane == 0 && word < Q.bitmap_words) bitmap[word] = (unsigned)mk;
    if (lane == 32 && word + 1u < Q.bitmap_words) bitmap[word + 1u] = (unsigned)(mk >> 32);
    if (keep) (Q.heavy_list + (size_t)Q.hv_cur * Q.heavy_cap)[idx] = t;
}

template <bool ROT, int SCL>
hipError_t launch_predict_variant(const RmParams &U, unsigned threshold, hipStream_t stream) {
    const unsigned blocks = (U.queue.n_tiles + 255u) / 256u;
    hipLaunchKernelGGL((rm_predict_kernel4<ROT, SCL>), dim3(blocks), dim3(256), 0, stream, U, threshold);
    return hipGetLastError();
}

}  // namespace

hipError_t rm_launch_predict(const RmParams &U, unsigned threshold, hipStream_t stream) {
    RM_DISPATCH_VARIANT(launch_predict_variant, U, U, threshold, stream);
}
