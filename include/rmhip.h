/*
 * rmhip.h -- C ABI of librmhip.so: the MI355X (gfx950) replacement for the
 * fragment-shader march of the Unity project symbiosinx/Raymarching.
 *
 * The reference drives its shader through Unity's Material property bag and a
 * full-screen blit:
 *     mat.SetMatrix("_FrustrumCorners", ...) ... Graphics.Blit(src, dest, mat)
 *                                      (Assets/Raymarching.cs:85-95)
 *     shadowMapMaterial.SetMatrix(...); Graphics.Blit(r, shadowMap, mat)
 *                                      (Assets/ShadowMap.cs:50-56)
 * This header is that boundary as plain C: an rm_ctx is one material bound to
 * one GPU, rm_set_* are Material.Set*, rm_blit* are Graphics.Blit.
 *
 * Conventions
 *  - Matrices are float[16] in Unity's Matrix4x4 field order (column-major:
 *    m00,m10,m20,m30,m01,...).  The shader reads M[i] as row i =
 *    {m[i], m[i+4], m[i+8], m[i+12]} (Assets/Raymarching.cginc:30,336-354).
 *  - Images are row-major with row 0 = BOTTOM row (blit uv origin is bottom-left,
 *    Assets/Raymarching.cginc:30 with Assets/Raymarching.cs:50-53).
 *  - Property names are the shader's, case-sensitive.  Unknown names are stored
 *    and ignored, as Unity does.
 *  - Every entry point returns 0 on success or a negative RM_E_* code; the
 *    message is available from rm_last_error().  Nothing throws or aborts.
 *  - A context is not re-entrant: one thread at a time, one context per GPU.
 */
#ifndef RMHIP_H
#define RMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_OK            0
#define RM_E_ARG        -1   /* null/invalid argument */
#define RM_E_DEVICE     -2   /* HIP runtime error (no GPU, launch failure, ...) */
#define RM_E_NOMEM      -3
#define RM_E_RANGE      -4   /* size beyond a documented limit */

#define RM_MAX_PRIMS   1024

typedef struct rm_ctx rm_ctx;

/* primitive kinds in the order scene() uses them (Assets/Raymarching.cginc:351-354);
 * torus is the reference's unused fifth primitive (cginc:179-182) */
enum { RM_SPHERE = 0, RM_BOX = 1, RM_LINK = 2, RM_CYLINDER = 3, RM_TORUS = 4 };

/* op bits of a primitive.  The scene is the pairwise tree of cginc:363 in index order; each of
 * its nodes has exactly one primitive that is the first one of its right operand, and that
 * primitive's RM_OP_SMAX bit makes the node smax (Assets/Raymarching.cginc:120-123) instead of
 * smin (:110-113).  RM_OP_NEGATE flips the sign of the primitive's distance (its complement), so
 * smax with a negated primitive carves it out of what stands to its left. */
enum { RM_OP_SMAX = 1, RM_OP_NEGATE = 2 };

/* One scene primitive for rm_set_primitives(): generalises the four rows of
 * _Positions/_Rotations/_Scales (Assets/Raymarching.cs:58-83) beyond the
 * reference's 4-slot limit (Assets/Raymarching.cs:60). */
typedef struct {
    int32_t type;       /* RM_* */
    int32_t op;         /* RM_OP_* bits, 0 = smooth union */
    float   pos[3];
    float   quat[4];    /* x,y,z,w */
    float   scale[3];
    float   color[3];
    float   params[4];  /* sphere r | box b.xyz | link le,r1,r2 | cylinder h,r | torus t.x,t.y */
} rm_primitive;

/* counters of the last blit / depth pass (all ranks of work the call covered) */
typedef struct {
    uint64_t n_pixels;       /* primary rays */
    uint64_t n_hit;          /* hit pixels = shadow rays */
    uint64_t sum_steps_p;    /* scene() evaluations of all primary marches */
    uint64_t sum_steps_s;    /* scene() evaluations of all shadow marches  */
} rm_stats;

/* ---- life cycle ---------------------------------------------------------
 * Replaces: the Material asset + its shader (Assets/Raymarching.mat,
 * Assets/ShadowMap.mat).  The context starts with the values serialized in
 * Assets/Raymarching.mat:62-95 and the default scene of
 * Assets/Scenes/SampleScene.unity.  No GPU call is made until the first
 * blit, so property-bag use works on a machine without a GPU. */
rm_ctx     *rm_create(int hip_device);
void        rm_destroy(rm_ctx *ctx);
const char *rm_last_error(rm_ctx *ctx);         /* ctx may be NULL: last creation error */
const char *rm_version(void);

/* ---- Material.SetFloat / SetInt / SetVector / SetMatrix -----------------
 * Replaces: mat.SetMatrix(name, Matrix4x4) at Assets/Raymarching.cs:86-90 and
 * Assets/ShadowMap.cs:52-53, and the inspector-serialized floats of
 * Assets/Raymarching.mat:62-95 declared at Assets/Raymarching.cginc:1-13 and
 * Assets/Raymarching.shader:30-46.  Unity built-ins the shader reads are explicit
 * properties here: _WorldSpaceCameraPos (shader:167), _WorldSpaceLightPos0
 * (shader:109,199), _ProjectionParams (shader:154).  _BackgroundColor (rgba in
 * [0,1]) stands in for _MainTex when rm_blit gets no source image. */
int rm_set_float (rm_ctx *ctx, const char *name, float value);
int rm_set_int   (rm_ctx *ctx, const char *name, int value);
int rm_set_vector(rm_ctx *ctx, const char *name, const float v[4]);
int rm_set_matrix(rm_ctx *ctx, const char *name, const float m[16]);
/* Material.GetFloat / GetInt / GetVector / GetMatrix; RM_E_ARG if never set */
int rm_get_float (rm_ctx *ctx, const char *name, float *value);
int rm_get_int   (rm_ctx *ctx, const char *name, int *value);
int rm_get_vector(rm_ctx *ctx, const char *name, float v[4]);
int rm_get_matrix(rm_ctx *ctx, const char *name, float m[16]);
/* Material.HasProperty: 1 if the march reads this name, 0 if it is only stored */
int rm_has_property(rm_ctx *ctx, const char *name);

/* Material.SetTexture for a texture that lives in HBM.  Replaces: mat.SetTexture("_ShadowMap", shadowMap.shadowMap),
 * the binding Assets/Raymarching.cs:92 keeps commented out, for the `sampler2D _ShadowMap` the shader declares at
 * Assets/Raymarching.shader:45.  d_texels: w x h fp32 depths as rm_shadowmap_render_device /
 * rm_shadowmap_directional_device write them (row 0 = bottom), caller-owned, alive while blits read it; NULL unbinds.
 * Other slot names are accepted and ignored (_MainTex and the camera depth texture are arguments of the blit).
 * The map is read when _RmShadowMode is 1 (perspective map; bind _ShadowMapVPMatrix = projection * worldToCamera of
 * the shadow camera, so that mul(_ShadowMapVPMatrix, float4(p,1)) is the clip-space position -- the commented line
 * Assets/Raymarching.cs:93 writes the two factors in the other order) or 2 (directional map; bind the rectangle as
 * _RmShadowMapCorners, rows TL,TR,BR,BL as Assets/ShadowMapDirectional.cs:11-16 orders them, and its
 * _RmShadowMapForward).  The lookup takes the place of the soft-shadow march in frag (Assets/Raymarching.shader:195):
 * 1 if the point is lit, 0 if the map holds something more than _RmShadowBias (relative, default 0.01) nearer. */
int rm_set_texture_device(rm_ctx *ctx, const char *name, const void *d_texels, int w, int h);

/* N-primitive scene (n = 0 returns to the 4-row matrices).  Extends
 * GetPositions/GetRotations/GetScales (Assets/Raymarching.cs:58-83); the tree is
 * the balanced pairwise smin of Assets/Raymarching.cginc:363 in index order (level by level,
 * an odd last element carried up), with smax nodes and negated primitives where the op bits
 * ask for them. */
int rm_set_primitives(rm_ctx *ctx, const rm_primitive *prims, int n);

/* ---- Graphics.Blit(src, dest, mat) --------------------------------------
 * Replaces: Assets/Raymarching.cs:94 + pass 0 of Assets/Raymarching.shader
 * (frag, :149-216).  Host-pointer form; synchronous.  src_rgba8 (the _MainTex
 * the shader shows on miss pixels) and scene_depth (linear eye depth, the
 * shader's Linear01Depth(..)*_ProjectionParams.z, :154) may be NULL.  Outputs:
 * RGBA8 colour (alpha 255), fp32 depth = r.length, optional 0/1 hit mask; any
 * may be NULL. */
int rm_blit(rm_ctx *ctx, const uint8_t *src_rgba8, const float *scene_depth,
            uint8_t *dst_rgba8, float *dst_depth, uint8_t *dst_hit, int w, int h);

/* rm_blit in two halves, for a caller that renders frame after frame into host memory (no counterpart in the reference,
 * whose Graphics.Blit is queued to Unity's render thread and never waited for by the script): rm_blit_async queues the
 * uploads of src / scene_depth, the march and the copies of the outputs back, and returns a ticket; rm_blit_wait(ticket)
 * returns once that call's outputs are in the caller's buffers, which hold exactly what rm_blit would have written.  Two
 * calls may be in flight: the copies back of one frame run beside the uploads and the march of the next (two staging sets in
 * HBM, each with a stream of its own).  A third call queues, in stream order, behind the one two calls earlier.  The
 * caller's buffers of a call belong to the library until rm_blit_wait for its ticket (or for a later one) has returned;
 * they should come from rm_host_alloc -- into pageable memory the copies are staged by the runtime and overlap little.
 * Properties are read when rm_blit_async is called.  rm_blit, the depth passes and rm_destroy wait for calls in flight. */
int rm_blit_async(rm_ctx *ctx, const uint8_t *src_rgba8, const float *scene_depth,
                  uint8_t *dst_rgba8, float *dst_depth, uint8_t *dst_hit, int w, int h, long long *ticket);
int rm_blit_wait(rm_ctx *ctx, long long ticket);

/* Page-locked host memory for the images of the host-pointer entry points.  No counterpart in the reference (Unity owns
 * its render targets).  rm_blit works with any host pointer; into pageable memory the runtime stages every copy (measured
 * 1.12 ms per 1920x1080 frame), into memory from rm_host_alloc the copies run at PCIe rate.  Free with rm_host_free
 * before rm_destroy.  Returns NULL on failure (rm_last_error has the text). */
void *rm_host_alloc(rm_ctx *ctx, size_t bytes);
void  rm_host_free(rm_ctx *ctx, void *p);

/* Same blit on device pointers, asynchronous on `hip_stream` (a hipStream_t; NULL
 * = the default stream).  This is the form a Unity native plugin uses with
 * Texture.GetNativeTexturePtr-style buffers already resident in HBM. */
int rm_blit_device(rm_ctx *ctx, const void *d_src_rgba8, const void *d_scene_depth,
                   void *d_dst_rgba8, void *d_dst_depth, void *d_dst_hit,
                   int w, int h, void *hip_stream);

/* The same blits with one more output: a per-pixel step-count plane, two uint16 per pixel (a 32-bit word:
 * low half = scene() evaluations of the primary march, i.e. what ray.steps counts at
 * Assets/Raymarching.cginc:37,387 plus the evaluation the loop broke on; high half = evaluations of the
 * soft-shadow march, 0 for miss pixels).  Replaces: the reference's only debug aid for the march, the
 * commented-out heat map `col += r.steps / 100 * _StepFactor` (Assets/Raymarching.shader:203), as data
 * instead of colour.  The frame sums of both halves are what rm_get_stats returns.  steps may be NULL. */
int rm_blit_steps(rm_ctx *ctx, const uint8_t *src_rgba8, const float *scene_depth,
                  uint8_t *dst_rgba8, float *dst_depth, uint8_t *dst_hit, uint16_t *dst_steps, int w, int h);
int rm_blit_steps_device(rm_ctx *ctx, const void *d_src_rgba8, const void *d_scene_depth,
                         void *d_dst_rgba8, void *d_dst_depth, void *d_dst_hit, void *d_dst_steps,
                         int w, int h, void *hip_stream);

/* rm_blit_device for callers that cannot allocate.  No counterpart in the reference: it is for code that must
 * not call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same arguments, same output bytes as rm_blit_device, but the device memory the kernels work in beyond the
 * caller's images (the persistent kernel's queue state, the counter slots, the reduced counters) is a static
 * device array of 557 568 bytes that every GPU the library runs on holds until the process ends, instead of
 * buffers the context allocates.  The call makes no hipMalloc/hipFree and creates no stream or event.
 * Limits: at most 32768 blocks of 16x16 pixels (ceil(w/16)*ceil(h/16), e.g. 3840x2160), else RM_E_RANGE; a
 * scene set with rm_set_primitives needs a table in allocated memory and is refused with RM_E_ARG.
 * All contexts on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads)
 * race; issue them in stream order.  The call leaves no event behind, so rm_get_stats after it waits on
 * hip_stream, which must still exist then; and the counters live in the shared array, so rm_get_stats and
 * rm_debug_counter_slots fail with RM_E_ARG once any later rm_blit_device_noalloc call on that GPU (of this or
 * another context) has reused it.
 * Under graph capture (hip_stream is capturing): the call records two kernel nodes, one that clears the tile-claim
 * counters and the frame's, both self-contained -- the launch reads and leaves no frame-to-frame state (the tile queue is
 * in image order, not ordered by the previous frame's costs), so the graph may be replayed any number of times,
 * interleaved with direct calls.  Make one direct call on the GPU first: the first call per GPU and per kernel
 * variant asks the runtime for the array's address and for occupancy figures.  If this context's previous call ran on
 * another stream, synchronize that stream before capturing (no host-side wait is possible during capture: RM_E_ARG). */
int rm_blit_device_noalloc(rm_ctx *ctx, const void *d_src_rgba8, const void *d_scene_depth,
                           void *d_dst_rgba8, void *d_dst_depth, void *d_dst_hit,
                           int w, int h, void *hip_stream);

/* Row-band shard of one frame for multi-GPU rendering: renders the rows y of the
 * w x h frame with (y / band_rows) % n_parts == part into COMPACT buffers that
 * hold only those rows, in increasing y.  rm_band_rows() gives their count.
 * src/scene_depth, if given, are full-frame device buffers. */
int rm_band_rows(int h, int band_rows, int part, int n_parts);
int rm_blit_bands_device(rm_ctx *ctx, const void *d_src_rgba8, const void *d_scene_depth,
                         void *d_dst_rgba8, void *d_dst_depth, void *d_dst_hit,
                         int w, int h, int band_rows, int part, int n_parts, void *hip_stream);

/* Batch of camera poses in one call: pose i uses frustum_corners[i*16..] and
 * cam_inv_view[i*16..] (both as rm_set_matrix takes them) and cam_pos[i*3..];
 * outputs are n_poses consecutive frames.  Replaces n_poses x (SetMatrix x2 +
 * Blit) of Assets/Raymarching.cs:86-94. */
int rm_blit_poses_device(rm_ctx *ctx, int n_poses, const float *frustum_corners,
                         const float *cam_inv_view, const float *cam_pos,
                         void *d_dst_rgba8, void *d_dst_depth, void *d_dst_hit,
                         int w, int h, void *hip_stream);

/* ---- ShadowMap.Render() --------------------------------------------------
 * Replaces: Assets/ShadowMap.cs:50-56 (SetMatrix x2 + Blit into `shadowMap`).
 * The reference's ShadowMap fragment is a colour-inverting stub
 * (Assets/ShadowMap.shader:20-25); here the pass is raymarch()
 * (Assets/Raymarching.cginc:367-392) from _WorldSpaceCameraPos along the rays
 * _FrustrumCorners/_CameraInvViewMatrix define, writing r.length. */
int rm_shadowmap_render(rm_ctx *ctx, float *dst_depth, int w, int h);
int rm_shadowmap_render_device(rm_ctx *ctx, void *d_dst_depth, void *d_dst_hit,
                               int w, int h, void *hip_stream);

/* ---- ShadowMapDirectional ------------------------------------------------
 * Replaces: the rectangle Assets/ShadowMapDirectional.cs:25-34 computes
 * (corners in its order TL,TR,BR,BL, :11-16) as an orthographic depth pass:
 * ray origin bilinear over the corners, direction `forward`, raymarch(). */
int rm_shadowmap_directional(rm_ctx *ctx, const float corners[12], const float forward[3],
                             float *dst_depth, int w, int h);
int rm_shadowmap_directional_device(rm_ctx *ctx, const float corners[12], const float forward[3],
                                    void *d_dst_depth, void *d_dst_hit,
                                    int w, int h, void *hip_stream);
/* Row-band shard of the same depth pass for multi-GPU rendering: the rows y of the w x h map over the rectangle of
 * Assets/ShadowMapDirectional.cs:25-34 with (y / band_rows) % n_parts == part, into COMPACT buffers that hold only
 * those rows in increasing y (rm_band_rows() of them), exactly as rm_blit_bands_device shards the frame.  The bands
 * of all parts put together are the map rm_shadowmap_directional_device writes, bit for bit. */
int rm_shadowmap_directional_bands_device(rm_ctx *ctx, const float corners[12], const float forward[3],
                                          void *d_dst_depth, void *d_dst_hit,
                                          int w, int h, int band_rows, int part, int n_parts, void *hip_stream);

/* ---- frame transport ------------------------------------------------------
 * No counterpart in the reference (one GPU, no transport).  The blit's alpha is the constant 255
 * (fixed4(col, 1), Assets/Raymarching.shader:214), so a finished frame that is gathered to the
 * displaying GPU over xGMI is sent as 3 bytes per pixel: rm_pack_rgb_device before the gather,
 * rm_unpack_rgb_device after it.  n_pixels must be a multiple of 4; asynchronous on hip_stream. */
int rm_pack_rgb_device(rm_ctx *ctx, const void *d_rgba8, void *d_rgb8, size_t n_pixels, void *hip_stream);
int rm_unpack_rgb_device(rm_ctx *ctx, const void *d_rgb8, void *d_rgba8, size_t n_pixels, void *hip_stream);

/* Lossless compact transport: a finished frame differs from what the displaying GPU already holds (the constant
 * _BackgroundColor, or the source image) only in the 8x8 tiles that contain a hit pixel.  rm_compact_frames_device
 * appends one record of rm_tile_record_bytes() (= 196: tile index = frame * tiles_per_frame + ty * tiles_x + tx as
 * uint32, then the tile's 64 pixels as RGB8, rows bottom-up) per such tile of the n_frames consecutive frames to
 * d_records, in no particular order, and leaves their number in the uint32 *d_count (device memory, cleared by the
 * call).  d_records must hold one record per tile in the worst case.  rm_expand_frames_device is the inverse on the
 * receiving GPU: every frame is first set to d_src_rgba8 (or, if NULL, to this context's _BackgroundColor), then the
 * records are written over it; the result equals the frames the blit wrote, byte for byte.  Both are asynchronous on
 * hip_stream. */
size_t rm_tile_record_bytes(void);
int rm_compact_frames_device(rm_ctx *ctx, const void *d_rgba8, const void *d_hit, int w, int h, int n_frames,
                             void *d_records, void *d_count, void *hip_stream);
int rm_expand_frames_device(rm_ctx *ctx, const void *d_records, uint32_t n_records, const void *d_src_rgba8,
                            void *d_dst_rgba8, int w, int h, int n_frames, void *hip_stream);
/* A displaying GPU that keeps its dense frames from one delivery to the next need not rewrite them whole:
 * rm_update_frames_device takes frames that an earlier rm_expand_frames_device / rm_update_frames_device call built
 * from d_old_records (same background), puts the background back into the tiles those records name and writes
 * d_new_records over the result.  The frames then equal what rm_expand_frames_device would build from d_new_records,
 * byte for byte, with only the tiles of the two record sets written. */
int rm_update_frames_device(rm_ctx *ctx, const void *d_old_records, uint32_t n_old, const void *d_new_records, uint32_t n_new,
                            const void *d_src_rgba8, void *d_dst_rgba8, int w, int h, int n_frames, void *hip_stream);

/* ---- counters -------------------------------------------------------------
 * No counterpart in the reference (its nearest thing is the commented-out
 * step-count heat map, Assets/Raymarching.shader:203).  rm_get_stats waits for
 * the stream of the last call and returns that call's counters. */
int rm_get_stats(rm_ctx *ctx, rm_stats *out);
/* What the exact culling of the table kernels saved in the last call (no counterpart in the reference): *visited = leaf
 * (primitive) evaluations the waves executed, *possible = the evaluations a walk of the whole table at every scene
 * evaluation would have executed, both summed over the waves of the call.  RM_E_ARG unless the last call ran the table
 * kernel with exact culling (tables of 33 to 512 primitives, _RmTableSkip not 0, none of frag's optional lines). */
int rm_get_leaf_stats(rm_ctx *ctx, uint64_t *visited, uint64_t *possible);
/* Developer aid, no counterpart in the reference: copies the raw per-workgroup counter
 * slots of the last call (words_per_slot 32-bit words each: hit pixels, primary
 * evaluations, shadow evaluations, then build-specific words) to host memory. */
int rm_debug_counter_slots(rm_ctx *ctx, uint32_t *out, int max_words, int *words_per_slot, int *n_slots);
/* Kernel variant for the next blits, no counterpart in the reference.  0 = automatic (the default).  1..3 and 5 are exact
 * kernels whose frames are bit-identical: 1 one wave per 8x8 tile, 2 persistent tile queue, 3 persistent waves with lane
 * refill, 5 four waves per tile that split the primitive table (tables of 3 to 512 primitives; others take 1; tables of 33 to 512
 * primitives, from 65 with op bits, also skip the subtrees that cannot change a bit of a march step's result, unless the
 * float property _RmTableSkip is 0; likewise the soft-shadow march of the unrotated 4-slot scene skips the sphere and the box
 * while they provably cannot change a bit of the step's result, unless the float property _RmPairSkip is 0).  4 = the
 * opt-in tolerance kernel for the reference's 4-slot scene (fused multiply-adds and hardware sqrt/rcp in the two march
 * loops): frames within the port's tolerance, not bit-identical; every other scene is rendered by the exact kernels. */
int rm_set_kernel(rm_ctx *ctx, int variant);
/* Tile order of the persistent kernel (no counterpart in the reference; none of this changes a pixel): a frame is rendered with
 * the tiles that were expensive in the context's previous frame first.  A frame without a usable previous one -- the first of a
 * context, another frame size, a camera cut, a primitive or the light moved far -- is preceded by a short pass that marches one
 * ray per 8x8 tile and supplies that order.  Float properties: _RmTileFeedback 0 renders every frame in the plain scattered
 * order, _RmPredictTiles 0 switches the pass off, _RmHeavyTileThreshold (default 24) and _RmPredictThreshold (default two thirds
 * of it) are the costs in scene evaluations from which a tile counts as expensive.  rm_get_float(ctx, "_RmPredictedFrames", &n)
 * reads how many frames were ordered by the pass so far (an error before the first). */

/* Unity's native-plugin loader calls these two, if a plugin exports them, when it loads and unloads the
 * library (Unity's IUnityInterface.h).  The scripts this library replaces (Assets/Raymarching.cs,
 * Assets/ShadowMap.cs, Assets/ShadowMapDirectional.cs) use nothing of the interface registry, so both do
 * nothing; they are exported so that the library behaves as a regular plugin under Assets/Plugins. */
void UnityPluginLoad(void *unity_interfaces);
void UnityPluginUnload(void);

#ifdef __cplusplus
}
#endif
#endif /* RMHIP_H */
