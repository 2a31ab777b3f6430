// rm_k_queue.hip -- persistent waves pulling 8x8 tiles from the ordered queue.
#include "rm_kernels_common.h"

namespace {

template <bool ROT, int SCL>
__global__ RM_QLB void rm_queue_kernel4(const RmParams U) {
    Scene4<ROT, SCL> sc(U);
    rm_queue_body(U, sc);
}

// the plain launch shapes (rm_queue_shape): the same kernel without the uniform branches of the other shapes
template <bool ROT, int SCL, int SHAPE>
__global__ RM_QLB void rm_queue_kernel4_plain(const RmParams U) {
    Scene4Plain<ROT, SCL, SHAPE> sc(U);
    rm_queue_body(U, sc);
}

// the table walk keeps four primitives' records in flight: 2 waves per SIMD (<= 256 VGPRs) avoids spills
template <bool ROT, int SCL>
__global__ __launch_bounds__(256, RM_Q_MIN_WAVES_ROT) void rm_queue_kernelN(const RmParams U) {
    if (U.lds_prims) {
        PrimTable<true> t; t.T = rm_stage_table(U, (U.queue.bitmap_words + 3u) & ~3u);
        SceneN<ROT, SCL, true> sc(t, U.n_prims, (U.scene_flags & RM_SF_CSG) != 0, U.tail_ops);
        rm_queue_body(U, sc);
    } else {
        PrimTable<false> t; t.T = U.prim_table;
        SceneN<ROT, SCL, false> sc(t, U.n_prims, (U.scene_flags & RM_SF_CSG) != 0, U.tail_ops);
        rm_queue_body(U, sc);
    }
}

template <bool ROT, int SCL>
hipError_t launch_queue_variant(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu) {
    const int shape = rm_queue_shape(U);
    if (blocks_per_cu) {
        int n4 = 0, nN = 0;
        const size_t lds4 = (size_t)U.queue.bitmap_words * 4;
        hipError_t e = shape == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, rm_queue_kernel4_plain<ROT, SCL, 1>, 256, lds4)
                     : shape == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, rm_queue_kernel4_plain<ROT, SCL, 3>, 256, lds4)
                     : shape == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, rm_queue_kernel4_plain<ROT, SCL, 2>, 256, lds4)
                     : shape == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, rm_queue_kernel4_plain<ROT, SCL, 4>, 256, lds4)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, rm_queue_kernel4<ROT, SCL>, 256, lds4);
        if (e != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nN, rm_queue_kernelN<ROT, SCL>, 256, (size_t)((U.queue.bitmap_words + 3u) & ~3u) * 4 + (size_t)U.lds_prims * sizeof(RmPrimDev));
        if (e != hipSuccess) return e;
        blocks_per_cu[0] = n4; blocks_per_cu[1] = nN;
        return hipSuccess;
    }
    const size_t lds = (size_t)((U.queue.bitmap_words + 3u) & ~3u) * 4 + (size_t)U.lds_prims * sizeof(RmPrimDev);
    if (U.n_prims > 0) hipLaunchKernelGGL((rm_queue_kernelN<ROT, SCL>), dim3(blocks), dim3(256), lds, stream, U);
    else if (shape == 1) hipLaunchKernelGGL((rm_queue_kernel4_plain<ROT, SCL, 1>), dim3(blocks), dim3(256), lds, stream, U);
    else if (shape == 2) hipLaunchKernelGGL((rm_queue_kernel4_plain<ROT, SCL, 2>), dim3(blocks), dim3(256), lds, stream, U);
    else if (shape == 3) hipLaunchKernelGGL((rm_queue_kernel4_plain<ROT, SCL, 3>), dim3(blocks), dim3(256), lds, stream, U);
    else if (shape == 4) hipLaunchKernelGGL((rm_queue_kernel4_plain<ROT, SCL, 4>), dim3(blocks), dim3(256), lds, stream, U);
    else               hipLaunchKernelGGL((rm_queue_kernel4<ROT, SCL>), dim3(blocks), dim3(256), lds, stream, U);
    return hipGetLastError();
}

}  // namespace

int rm_queue_shape(const RmParams &U) {
    const bool base = U.body == RM_BODY_PRIMS && U.n_prims == 0 && U.mode == RM_MODE_FRAG && U.n_parts == 1 &&
                      U.dst_rgba && U.dst_depth && U.dst_hit && U.inv_w != 0.0f && U.inv_h != 0.0f;
    if (!base) return 0;
    const bool single = !U.views && U.n_poses == 1;
    if (U.shadow_mode != 0) return (single && !U.src && !U.sdepth && U.shadow_map) ? 4 : 0;
    if (!U.src && !U.sdepth) return single ? 1 : (U.views ? 2 : 0);
    if (U.src && U.sdepth && single) return 3;
    return 0;
}

hipError_t rm_launch_queue(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu) {
    if (U.body != RM_BODY_PRIMS) return rm_launch_queue_body(U, blocks, stream, blocks_per_cu);
    RM_DISPATCH_VARIANT(launch_queue_variant, U, U, blocks, stream, blocks_per_cu);
}
