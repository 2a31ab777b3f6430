// rm_k_stream.hip -- persistent waves with lane-level ray compaction (experimental).
#include "rm_kernels_common.h"
#include "rm_coop.h"

namespace {

template <bool ROT, int SCL>
__global__ __launch_bounds__(256, RM_Q_MIN_WAVES_ROT) void rm_stream_kernelN(const RmParams U) {
    if (U.lds_prims) {
        PrimTable<true> t; t.T = rm_stage_table(U, 0);
        SceneN<ROT, SCL, true> sc(t, U.n_prims, (U.scene_flags & RM_SF_CSG) != 0, U.tail_ops);
        rm_stream_body(U, sc);
    } else {
        PrimTable<false> t; t.T = U.prim_table;
        SceneN<ROT, SCL, false> sc(t, U.n_prims, (U.scene_flags & RM_SF_CSG) != 0, U.tail_ops);
        rm_stream_body(U, sc);
    }
}

// big tables staged in LDS: the workgroup's four waves carry the same 64 rays and split the table (rm_coop.h), so a lane
// that is refilled keeps all four waves busy and the longest ray is a four times shorter chain
template <bool ROT, int SCL, bool CSG>
__global__ __launch_bounds__(256, 3) void rm_stream_kernelN_coop(const RmParams U) {
    __shared__ float s_xw[2 * 4 * 64];
    __shared__ float s_x4[4 * 4 * 64];
    PrimTable<true> t;
    t.T = rm_stage_table(U, 0);
    const SceneNCoop<ROT, SCL> sc = rm_make_coop_scene<ROT, SCL>(U, t, (RM_LDS float *)s_xw, (RM_LDS float *)s_x4, CSG);
    rm_stream_body<true>(U, sc);
}

template <bool ROT, int SCL>
__global__ RM_QLB void rm_stream_kernel4(const RmParams U) {
    Scene4<ROT, SCL> sc(U);
    rm_stream_body(U, sc);
}

template <bool ROT, int SCL>
hipError_t launch_stream_variant(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu) {
    const size_t lds = (size_t)U.lds_prims * sizeof(RmPrimDev);
    const bool coop = rm_stream_is_coop(U), csg = (U.scene_flags & RM_SF_CSG) != 0;
    if (blocks_per_cu) {
        int n4 = 0, nN = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n4, rm_stream_kernel4<ROT, SCL>, 256, 0);
        if (e != hipSuccess) return e;
        if (coop && csg)  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nN, rm_stream_kernelN_coop<ROT, SCL, true>, 256, lds);
        else if (coop)    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nN, rm_stream_kernelN_coop<ROT, SCL, false>, 256, lds);
        else              e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nN, rm_stream_kernelN<ROT, SCL>, 256, lds);
        if (e != hipSuccess) return e;
        blocks_per_cu[0] = n4; blocks_per_cu[1] = nN;
        return hipSuccess;
    }
    if (coop && csg)        hipLaunchKernelGGL((rm_stream_kernelN_coop<ROT, SCL, true>), dim3(blocks), dim3(256), lds, stream, U);
    else if (coop)          hipLaunchKernelGGL((rm_stream_kernelN_coop<ROT, SCL, false>), dim3(blocks), dim3(256), lds, stream, U);
    else if (U.n_prims > 0) hipLaunchKernelGGL((rm_stream_kernelN<ROT, SCL>), dim3(blocks), dim3(256), lds, stream, U);
    else                    hipLaunchKernelGGL((rm_stream_kernel4<ROT, SCL>), dim3(blocks), dim3(256), 0, stream, U);
    return hipGetLastError();
}

}  // namespace

bool rm_stream_is_coop(const RmParams &U) {
    return U.body == RM_BODY_PRIMS && U.n_prims >= RM_COOP_MIN_PRIMS && U.lds_prims == U.n_prims;
}

hipError_t rm_launch_stream(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu) {
    RM_DISPATCH_VARIANT(launch_stream_variant, U, U, blocks, stream, blocks_per_cu);
}
