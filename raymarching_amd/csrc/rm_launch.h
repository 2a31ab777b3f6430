// rm_launch.h -- host-callable launchers implemented in the rm_k_*.hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "rm_device.h"

// kernel variants selectable through rm_set_kernel()
enum { RM_KERNEL_AUTO = 0, RM_KERNEL_TILE = 1, RM_KERNEL_QUEUE = 2, RM_KERNEL_STREAM = 3, RM_KERNEL_FAST = 4, RM_KERNEL_COOP = 5 };

// one-thread-per-pixel tile kernel over U.w x U.rows x U.n_poses
hipError_t rm_launch_tiles(const RmParams &U, hipStream_t stream);
// the same with frag's optional lines (U.fx_flags / U.fx_view); called by rm_launch_tiles with its grid
hipError_t rm_launch_tiles_fx(const RmParams &U, unsigned grid_x, unsigned grid_y, unsigned grid_z, hipStream_t stream);

// the same two launchers for the scene() bodies other than the primitives (U.body != RM_BODY_PRIMS); called by rm_launch_tiles
// and rm_launch_queue
hipError_t rm_launch_tiles_body(const RmParams &U, unsigned grid_x, unsigned grid_y, unsigned grid_z, hipStream_t stream);
hipError_t rm_launch_queue_body(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu);

// 32-bit words per counter slot
unsigned rm_slot_words();
// number of blocks (= counter slots) rm_launch_tiles uses for a launch of this shape
unsigned rm_tile_blocks(int w, int rows, int n_poses);
// sums n_blocks counter slots into out[0..2] (device memory)
// RGBA8 -> RGB8 (unpack false: first argument in, second out) or RGB8 -> RGBA8 with alpha 255 (unpack true:
// first argument is the RGB8 input, second the RGBA8 output); n_pixels must be a multiple of 4
hipError_t rm_pack_rgb(const void *in, void *out, size_t n_pixels, bool unpack, hipStream_t stream);
// Frame transport.  rm_compact_tiles appends one record of rm_record_words() dwords (tile index, 64 pixels as RGB8) per
// 8x8 tile of the n_frames frames that holds a hit pixel and counts them in *counter (cleared first); rm_expand_tiles
// writes such records back over frames that already hold the background.
unsigned rm_record_words();
hipError_t rm_compact_tiles(const void *rgba, const void *hit, int w, int h, unsigned n_frames, void *records, void *counter, hipStream_t stream);
hipError_t rm_expand_tiles(const void *records, unsigned n_records, int w, int h, unsigned n_frames, void *rgba, hipStream_t stream);
hipError_t rm_restore_tiles(const void *records, unsigned n_records, int w, int h, unsigned n_frames, const void *src_rgba8, uint32_t bg8,
                            void *rgba, hipStream_t stream);
hipError_t rm_fill_from_src(const void *src_rgba8, void *dst_rgba8, size_t pixels_per_frame, unsigned n_frames, hipStream_t stream);
// clears n_words 32-bit words of device memory with a kernel (a kernel node when recorded into a graph)
hipError_t rm_zero_words(void *p, unsigned n_words, hipStream_t stream);
hipError_t rm_reduce_stats(const unsigned *partials, unsigned n_blocks, unsigned long long *out, hipStream_t stream);

// persistent tile-queue kernel: `blocks` workgroups of 256 threads.  With blocks_per_cu != nullptr
// nothing is launched; blocks_per_cu[0..1] receive the resident blocks per CU of the 4-slot and
// the table variant that U selects.
hipError_t rm_launch_queue(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu);
// the instantiation rm_launch_queue picks for U: 1 = the plain frame (4-slot scene, fragment pass, whole frame, one pose with
// the view in the arguments, no source image or depth texture, all three output planes), 2 = the same for a batch of poses
// with their views in device memory, 3 = as 1 but with source image and depth texture (the reference's Blit), 4 = as 1 with the
// shadow term looked up in the bound _ShadowMap (U.shadow_mode != 0) instead of marched, 0 = the general kernel (which has no
// lookup: with U.shadow_mode != 0 only shape 4 may be launched)
int rm_queue_shape(const RmParams &U);
// Tile costs without a previous frame (4-slot scenes): marches the centre ray of every 8x8 tile of the launch U describes until
// it has cost `threshold` scene evaluations and lists the tiles whose ray got there in heavy-tile set U.queue.hv_cur (which must
// be clear), as the previous frame of rm_launch_queue would have
hipError_t rm_launch_predict(const RmParams &U, unsigned threshold, hipStream_t stream);
int rm_variant_index(const RmParams &U);

// persistent kernel with lane-level ray compaction; same calling convention as rm_launch_queue
hipError_t rm_launch_stream(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu);
// true if rm_launch_stream runs U with one workgroup (not one wave) per tile: big tables staged whole in LDS, whose
// evaluations the four waves of a workgroup share
bool rm_stream_is_coop(const RmParams &U);

// the persistent kernel with the march loops on the fast (tolerance, not bit-exact) distance evaluation; 4-slot scenes
// only; same calling convention as rm_launch_queue
hipError_t rm_launch_fast(const RmParams &U, unsigned blocks, hipStream_t stream, int *blocks_per_cu);

// exact kernel for table scenes: one block per 8x8 tile, its four waves share the tile's rays and split the table
// (tables of at least RM_COOP_MIN_PRIMS primitives that are staged whole in LDS; 3 is the smallest table that splits into
// three chunks, and measured on 1080p frames the shared tile is 1.2-2.3x faster than one wave per tile from there on)
#ifndef RM_COOP_MIN_PRIMS
#define RM_COOP_MIN_PRIMS 3
#endif
unsigned rm_coop_blocks(int w, int rows, int n_poses);
hipError_t rm_launch_coop_tiles(const RmParams &U, hipStream_t stream);
// the same kernel with exact culling in the two marches and the normal taps (rm_skip.h): tables of 33 to 512 primitives, 65
// to 512 with op bits; rm_skip_applies() says whether U is one
bool rm_skip_applies(const RmParams &U);
hipError_t rm_launch_coop_skip_tiles(const RmParams &U, hipStream_t stream);
