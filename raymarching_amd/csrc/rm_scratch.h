// rm_scratch.h -- layout of the per-GPU static scratch array (rm_scratch.hip) that the entry point which
// allocates nothing works in: the persistent kernel's queue state, the counter slots and the reduced counters.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "rm_device.h"

#define RM_SCRATCH_MAX_DEVICES 64
// one heavy-tile bitmap: 160 K tiles of 8x8 pixels (a 3840x2160 frame has 129 600)
#define RM_SCRATCH_BM_WORDS    5120u
// queue state: claim counters and heavy counts (1088 words with 8 counters per set), three heavy lists of 8192 entries,
// three bitmaps
#define RM_SCRATCH_QUEUE_BYTES ((size_t)(2u * RM_Q_SHARDS * RM_Q_STRIDE + 64u + 3u * 8192u + 3u * RM_SCRATCH_BM_WORDS) * 4u)
// counter slots: 32768 slots of RM_SLOT_WORDS words (three, unless this is a timeline build) = one slot per 16x16-pixel
// block of a 3840x2160 frame (32 400)
#define RM_SCRATCH_SLOT_BYTES  ((size_t)32768u * 4u * RM_SLOT_WORDS)
#define RM_SCRATCH_STATS_BYTES ((size_t)256u)
#define RM_SCRATCH_BYTES       (RM_SCRATCH_QUEUE_BYTES + RM_SCRATCH_SLOT_BYTES + RM_SCRATCH_STATS_BYTES)

// Device address of this GPU's copy of the scratch array.  `device` must be the current device.  The first call
// per device asks the runtime for the address; later calls return the stored one.
hipError_t rm_scratch_address(int device, void **ptr);
