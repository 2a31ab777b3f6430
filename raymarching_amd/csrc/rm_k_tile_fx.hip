// rm_k_tile_fx.hip -- the tile kernels with frag's optional lines (rm_compute_pixel_fx) and the domain
// operators in front of scene() (SceneDom).
#include "rm_kernels_common.h"

namespace {

template <bool ROT, int SCL>
__global__ RM_LB void rm_tile_kernel4_fx(const RmParams U) {
    Scene4<ROT, SCL> sc(U);
    rm_pixel<true>(U, SceneDom<Scene4<ROT, SCL>>(sc, U));
}

template <bool ROT, int SCL>
__global__ RM_LB void rm_tile_kernelN_fx(const RmParams U) {
    if (U.lds_prims) {
        PrimTable<true> t; t.T = rm_stage_table(U, 0);
        SceneN<ROT, SCL, true> sc(t, U.n_prims, (U.scene_flags & RM_SF_CSG) != 0, U.tail_ops);
        rm_pixel<true>(U, SceneDom<SceneN<ROT, SCL, true>>(sc, U));
    } else {
        PrimTable<false> t; t.T = U.prim_table;
        SceneN<ROT, SCL, false> sc(t, U.n_prims, (U.scene_flags & RM_SF_CSG) != 0, U.tail_ops);
        rm_pixel<true>(U, SceneDom<SceneN<ROT, SCL, false>>(sc, U));
    }
}

__global__ RM_LB void rm_tile_kernelF_fx(const RmParams U) {
    SceneF sc(U);
    rm_pixel<true>(U, SceneDom<SceneF>(sc, U));
}
template <bool ROT, int SCL>
hipError_t launch_variant_fx(const RmParams &U, dim3 grid, hipStream_t stream) {
    if (U.n_prims > 0) hipLaunchKernelGGL((rm_tile_kernelN_fx<ROT, SCL>), grid, dim3(RM_BLOCK), (size_t)U.lds_prims * sizeof(RmPrimDev), stream, U);
    else               hipLaunchKernelGGL((rm_tile_kernel4_fx<ROT, SCL>), grid, dim3(RM_BLOCK), 0, stream, U);
    return hipGetLastError();
}

}  // namespace

hipError_t rm_launch_tiles_fx(const RmParams &U, unsigned gx, unsigned gy, unsigned gz, hipStream_t stream) {
    const dim3 grid(gx, gy, gz);
    if (U.body != RM_BODY_PRIMS) {
        hipLaunchKernelGGL(rm_tile_kernelF_fx, grid, dim3(RM_BLOCK), 0, stream, U);
        return hipGetLastError();
    }
    RM_DISPATCH_VARIANT(launch_variant_fx, U, U, grid, stream);
}
