// rm_k_fast.hip -- the opt-in tolerance kernel for the 4-slot scene: the persistent tile-queue kernel with the march loops
// on the fast distance evaluation of rm_fast.h.  Never selected automatically.
#include "rm_kernels_common.h"
#include "rm_fast.h"

namespace {

// SHAPE: the launch shape promised at compile time (rm_queue_shape): 0 any, 1 the plain frame, 2 a plain pose batch
template <bool ROT, int SCL, int SHAPE>
__global__ RM_QLB void rm_queue_kernel4_fast(const RmParams U) {
    rmf::Scene4Fast<ROT, SCL, SHAPE> sc(U);
    rm_queue_body(U, sc);
}

template <bool ROT, int SCL, int SHAPE>
hipError_t launch_fast_shape(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu) {
    const size_t lds = (size_t)((U.queue.bitmap_words + 3u) & ~3u) * 4;
    if (blocks_per_cu) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rm_queue_kernel4_fast<ROT, SCL, SHAPE>, 256, lds);
        blocks_per_cu[0] = blocks_per_cu[1] = n;
        return e;
    }
    hipLaunchKernelGGL((rm_queue_kernel4_fast<ROT, SCL, SHAPE>), dim3(blocks), dim3(256), lds, stream, U);
    return hipGetLastError();
}

template <bool ROT, int SCL>
hipError_t launch_fast_variant(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu) {
    const int shape = rm_queue_shape(U);
    if (shape == 1) return launch_fast_shape<ROT, SCL, 1>(U, blocks, stream, blocks_per_cu);
    if (shape == 2) return launch_fast_shape<ROT, SCL, 2>(U, blocks, stream, blocks_per_cu);
    if (shape == 3) return launch_fast_shape<ROT, SCL, 3>(U, blocks, stream, blocks_per_cu);
    // (shape 4, the shadow-map lookup, is never routed here)
    return launch_fast_shape<ROT, SCL, 0>(U, blocks, stream, blocks_per_cu);
}

}  // namespace

// 4-slot scenes only (U.n_prims == 0, U.body == RM_BODY_PRIMS); a scene that needs a true division by its scale
// runs the reciprocal variant here
hipError_t rm_launch_fast(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu) {
    const bool rot = !(U.scene_flags & RM_SF_IDENT_ROT), unit = (U.scene_flags & RM_SF_UNIT_SCALE) != 0;
    if (rot) return unit ? launch_fast_variant<true, RM_SCL_UNIT>(U, blocks, stream, blocks_per_cu)
                         : launch_fast_variant<true, RM_SCL_RCP>(U, blocks, stream, blocks_per_cu);
    return unit ? launch_fast_variant<false, RM_SCL_UNIT>(U, blocks, stream, blocks_per_cu)
                : launch_fast_variant<false, RM_SCL_RCP>(U, blocks, stream, blocks_per_cu);
}

