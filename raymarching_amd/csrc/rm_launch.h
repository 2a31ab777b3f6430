// rm_launch.h -- host-callable launchers implemented in rm_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "rm_device.h"

// kernel variants selectable through rm_set_kernel()
enum { RM_KERNEL_AUTO = 0, RM_KERNEL_TILE = 1 };

// one-thread-per-pixel tile kernel over U.w x U.rows x U.n_poses
hipError_t rm_launch_tiles(const RmParams &U, hipStream_t stream);
