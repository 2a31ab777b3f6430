// rm_device.h -- plain-data structs shared by the host side of librmhip and its
// gfx950 kernels.  Everything the march needs per launch travels in RmParams as
// a by-value kernel argument, so wave-uniform scene constants are read through
// the scalar unit (constant cache -> SGPRs) rather than per lane.
#pragma once
#include <stdint.h>

// One primitive with every uniform-only sub-expression of the reference's
// scene() wrapper evaluated once on the host, in the same fp32 operation order
// the shader would use (Assets/Raymarching.cginc:62-66, 336-354):
//   u  = -q.xyz                      c  = q.w*q.w - dot(u,u)
//   w2 = 2*q.w                       ms = min(scale.x, min(scale.y, scale.z))
struct RmPrimDev {
    // Field order = order of need: a kernel variant reads the record from LDS in 16-byte quads and
    // the compiler drops the quads a variant never touches (unit scale: no quad 3/4; no rotation:
    // no quad 1; distance-only evaluation: no quad 5).
    float px, py, pz;      // _Positions row
    int32_t type;          // bits 0-7: primitive kind.  In records 4g, bit 8+l: the merge of group g's
                           // finished subtree with stack level l is smax (see SceneN::tree)
    float ux, uy, uz;      // -q.xyz
    float c;               // q.w*q.w - dot(u,u)
    float w2;              // 2*q.w
    float a0, a1, a2;      // sphere: r*.5 | box: b*.5 | link: le,r1,r2 | cylinder: h,r | torus: t.x,t.y
    float sx, sy, sz;      // scale
    float ms;              // min(scale)
    float rx, ry, rz;      // RN(1/scale), used when RM_PF_RCP is set
    float nsgn;            // -1 if the distance is negated (complement solid), else +1
    float cr, cg, cb;      // colour assigned after the primitive call (cginc:356-359)
    float osgn;            // -1 if the node whose right operand starts at this primitive is smax, else +1
};
static_assert(sizeof(RmPrimDev) == 96, "RmPrimDev layout");

enum : int32_t {
    RM_PF_IDENT_ROT  = 1,   // q == (0,0,0,1): rotate() returns its argument bit for bit
    RM_PF_UNIT_SCALE = 2,   // scale == (1,1,1): p/scale == p, d*min(scale) == d
    RM_PF_CSG        = 8,   // carries an op bit (host-side bookkeeping only)
    RM_PF_RCP        = 4,   // p/s may be computed from RN(1/s) with one exact residual correction (see rm_div_by)
};

// scene-wide specialisation, the AND of the per-primitive flags
enum : int32_t {
    RM_SF_IDENT_ROT  = 1,
    RM_SF_UNIT_SCALE = 2,
    RM_SF_RCP        = 4,
    RM_SF_CSG        = 8,   // some primitive carries an op bit (smax node or negated distance)
    RM_SF_LIPSCHITZ  = 16,  // 4-slot scene: every scale is finite and positive, so every primitive's distance (and any smin of
                            // them) changes by at most the distance the sample point moves
};

#define RM_TYPE_MASK 0xff
#define RM_CASCADE_SHIFT 8

#define RM_LDS_PRIMS 512       // largest primitive table staged in LDS (48 KB)

struct RmView {
    float ro[3];           // _WorldSpaceCameraPos
    float bl[3], br[3], tl[3], tr[3];   // world-space corner rays: mul((float3x3)_CameraInvViewMatrix, _FrustrumCorners[i].xyz)
    float pad;
};
static_assert(sizeof(RmView) == 64, "RmView layout");

// Tile queue of the persistent kernel and its frame-to-frame state (all device memory owned
// by the context).  Three heavy-list/bitmap sets and two counter sets rotate so that a frame
// never clears anything it, or a concurrently running wave of the same frame, still reads.
// pixel tile one wave renders at a time in the persistent kernel: RM_TW x RM_TH = 64
#ifndef RM_TW
#define RM_TW 8
#endif
#define RM_TH (64 / RM_TW)
#ifndef RM_Q_SHARDS
#define RM_Q_SHARDS 8
#endif
#ifndef RM_Q_CHUNK
#define RM_Q_CHUNK 4          // queue entries per dynamic claim
#endif
#ifndef RM_Q_STRIDE
#define RM_Q_STRIDE 64
#endif
//       // words between shard counters: one counter per 256-B line, so shards hit different L2 channels
#define RM_Q_SOLO 0xffffffffu
struct RmQueue {
    uint32_t *counters;      // [2][RM_Q_SHARDS][RM_Q_STRIDE] claim counters (first word of each stride)
    uint32_t *heavy_count;   // [3]
    uint32_t *heavy_list;    // [3][heavy_cap] tile ids that were expensive in the previous frame
    uint32_t *bitmaps;       // [3][bitmap_words] one bit per tile: set = on the heavy list of that set
    uint32_t  n_tiles, tiles_x, tiles_y;
    uint32_t  bitmap_words;  // ceil(n_tiles / 32), or 0 when no history is kept
    uint32_t  heavy_cap, heavy_threshold;
    uint32_t  scatter_mul;       // coprime with n_tiles
    uint32_t  static_per_wave;   // entries every wave without a heavy tile takes without claiming
    uint32_t  use_history;   // 0: previous frame unusable (first frame, size change)
    uint32_t  cnt_cur, cnt_clr;          // counter sets: in use / cleared for the next frame.  cnt_clr == RM_Q_SOLO: a self-contained
                                         // launch (recorded into a graph): no history read or appended, no set cleared
    uint32_t  hv_cur, hv_nxt, hv_clr;    // heavy sets: read / appended / cleared
    // t / tiles_x for t < 2^24 as (t * tx_magic) >> 37, exact for tiles_x <= 4096 (tx_magic = ceil(2^37 / tiles_x))
    uint64_t  tx_magic;
};

// words per wave / block in the counter-slot buffer; a timeline build (developer tool) appends start/end timestamps
// (100 MHz s_memrealtime), the hardware id and the wave's last four tiles
#ifdef RM_TIMELINE
#define RM_SLOT_WORDS 16
#else
#define RM_SLOT_WORDS 3
#endif

enum : int32_t { RM_MODE_FRAG = 0, RM_MODE_DEPTH_PERSP = 1, RM_MODE_DEPTH_ORTHO = 2 };

struct RmParams {
    RmView   view;             // used when views == nullptr
    float    light[3];         // _WorldSpaceLightPos0.xyz
    int32_t  n_steps;          // iterations of `for (i = 0; (float)i < min(2000,_MaxSteps); i++)` (cginc:369,372; shader:104,108)
    float    max_dist;         // _MaxDist
    float    shadow_depth;     // min(2000, _MaxDist)           (shader:103)
    float    far_plane;        // _ProjectionParams.z
    float    step_factor;      // _StepFactor
    float    nss;              // max(0, _NormalSampleScale)    (shader:51)
    float    nss_raw;          // _NormalSampleScale            (shader:126)
    float    bg[3];            // constant _MainTex colour
    uint32_t bg_rgba8;         // the same, quantised, alpha 255
    float    ortho_c[4][3];    // directional pass: corners TL,TR,BR,BL
    float    ortho_fwd[3];
    int32_t  w, h;             // full frame size
    float    inv_w, inv_h;     // RN(1/w), RN(1/h) when every pixel-centre coordinate divides exactly through rm_div_by, else 0
    int32_t  rows;             // rows rendered by this launch (compact output rows)
    int32_t  band_rows, part, n_parts;
    int32_t  mode;
    int32_t  n_poses;
    int32_t  n_prims;          // 0: prims4 below
    int32_t  lds_prims;        // n_prims if the table is staged in LDS by each workgroup, else 0
    int32_t  scene_flags;
    uint32_t tail_ops;       // bit l: the final merge with stack level l is smax (see SceneN::tree)
    const RmView    *views;    // n_poses views, or nullptr
    const RmPrimDev *prim_table;
    const uint8_t   *src;      // full-frame RGBA8 or nullptr
    const float     *sdepth;   // full-frame linear eye depth or nullptr
    uint8_t  *dst_rgba;
    float    *dst_depth;
    uint8_t  *dst_hit;
    uint32_t *dst_steps;       // optional, tile kernels only: per pixel, scene() evaluations of the primary march (low 16 bits,
                               // ray.steps of cginc:37,387 counted in evaluations) and of the shadow march (high 16 bits)
    uint32_t *partials;        // counter slots (per block or per wave): n_hit, primary evals, shadow evals
    uint32_t *leaf_stats;      // optional, exact-culling table kernel only: per wave two words, leaf evaluations executed and
                               // leaf evaluations a full table walk would have executed
    RmQueue   queue;           // persistent kernel only
    RmPrimDev prims4[4];
    // fractal scene() bodies (cginc:264-313); the wrapper transform is prims4[0]
    int32_t  body;             // RM_BODY_*
    float    f_scale;          // _FractalScale
    float    f_scale_m1;       // _FractalScale - 1      (cginc:284-286)
    float    f_rot[6];         // cos,sin of _FractalRotationX, Y, Z (cginc:53-54)
    float    f_pow3;           // pow(3.0, -12)          (cginc:312)
    float    f_pows[16];       // pow(_FractalScale, -i) (cginc:292)
    float    f_power;          // e of mandelbulb (cginc:194) / mandelbrot (cginc:229)
    float    f_rot_time[6];    // composite body: cos,sin of float3(0, 0, _Time.y)                                 (cginc:349)
    float    f_rot_sponge[6];  // composite body: cos,sin of float3(_Time.y*.25, _Time.y*1.4*.25, _Time.y*1.6*.25) (cginc:349)
    // lines of frag the reference keeps commented out (0 = the shader as it stands); tile kernels only
    uint32_t fx_flags;         // RM_FX_*
    int32_t  light_model;      // 0 dot(normal, _WorldSpaceLightPos0)*.5+.5 (shader:199), 1 getbrightness (shader:92-100), 2 getbrightnesshard (:84-90)
    float    light_pos[3];     // _LightPosition: position of the `light l` frag builds (shader:157-162)
    float    refr_index;       // _RefractiveIndex, the eta of refract2 (shader:142-147)
    int32_t  fx_view;          // 1 shadow (shader:200), 2 normal*.5+.5 (:201), 3 clamp(scene(hitpoint).r*2.5-1.75,0,1) (:198)
    // domain operator applied to p at the top of scene() (functions the reference defines and never calls);
    // tile kernels with the optional lines only
    // shadow-map lookup in place of the shadow march (tile kernels with the optional lines only): the reference declares
    // _ShadowMap and _ShadowMapVPMatrix (shader:45-46) and never reads them
    int32_t  shadow_mode;      // 0 march (shader:102-123), 1 perspective map, 2 directional map
    int32_t  sm_w, sm_h;
    const float *shadow_map;   // sm_w x sm_h depths written by a depth pass (r.length)
    float    sm_row0[4], sm_row1[4], sm_row3[4];   // rows x, y, w of _ShadowMapVPMatrix
    float    sm_bl[3], sm_e1[3], sm_e2[3], sm_fwd[3];   // directional: BL, BR - BL, TL - BL, forward
    float    sm_n1, sm_n2;     // dot(e1,e1), dot(e2,e2)
    float    sm_onebias;       // 1 + _RmShadowBias
    int32_t  dom_op;           // RM_DOM_*
    float    dom_period[3];    // m of shmod (cginc:94-99) per axis; <= 0: axis left alone
    float    dom_size;         // b of boxfold (cginc:156) / R of spherefold (cginc:161)
};

enum : int32_t { RM_DOM_NONE = 0, RM_DOM_REPEAT = 1, RM_DOM_BOXFOLD = 2, RM_DOM_SPHEREFOLD = 3 };

enum : uint32_t { RM_FX_STEP_HEAT = 1, RM_FX_VOLUMETRIC = 2, RM_FX_BOUNCE = 4, RM_FX_SCATTER = 8, RM_FX_REFRACT2 = 16 };

enum : int32_t { RM_BODY_PRIMS = 0, RM_BODY_MENGER = 1, RM_BODY_SIERPINSKI = 2, RM_BODY_MANDELBULB = 3, RM_BODY_MANDELBROT = 4, RM_BODY_TERRAIN = 5,
                 RM_BODY_COMPOSITE = 6 };   // the commented line cginc:349: a mandelbulb whose exponent is a menger distance + 1.1
