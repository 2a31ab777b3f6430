// rm_k_body.hip -- the scene() bodies other than the primitives (menger, sierpinski, mandelbulb, mandelbrot, terrain):
// tile kernel and persistent queue kernel.  A translation unit of its own because it is compiled without the SLP
// vectoriser: the folds and rotations of these bodies pack into fp32 pairs that need more register moves than they save
// (menger close-up at 1080p: 1.24 ms with, 0.93 ms without; terrain 1.56 -> 1.45 ms), while the 4-slot kernels gain 7 % from
// the same packing.
#include "rm_kernels_common.h"

namespace {

// the slot-0 transform is specialised by wave-uniform branches
__global__ RM_LB void rm_tile_kernelF(const RmParams U) {
    SceneF sc(U);
    rm_pixel(U, sc);
}

__global__ __launch_bounds__(256, RM_Q_MIN_WAVES_ROT) void rm_queue_kernelF(const RmParams U) {
    SceneF sc(U);
    rm_queue_body(U, sc);
}

}  // namespace

hipError_t rm_launch_tiles_body(const RmParams &U, unsigned grid_x, unsigned grid_y, unsigned grid_z, hipStream_t stream) {
    hipLaunchKernelGGL(rm_tile_kernelF, dim3(grid_x, grid_y, grid_z), dim3(RM_BLOCK), 0, stream, U);
    return hipGetLastError();
}

hipError_t rm_launch_queue_body(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu) {
    const size_t lds = (size_t)U.queue.bitmap_words * 4;
    if (blocks_per_cu) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rm_queue_kernelF, 256, lds);
        blocks_per_cu[0] = blocks_per_cu[1] = n;
        return e;
    }
    hipLaunchKernelGGL(rm_queue_kernelF, dim3(blocks), dim3(256), lds, stream, U);
    return hipGetLastError();
}
