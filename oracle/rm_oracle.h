/*
 * rm_oracle.h -- CPU oracle for the sphere-tracing path of symbiosinx/Raymarching.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker that the HIP path is compared
 * against; nothing in the shipped library links, loads or calls it.  Only the
 * test-suite, the smoke check and the benchmark's CPU-baseline leg may use it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference holds no golden images, tests
 * or known-answer vectors and only runs inside the Unity editor, so this oracle
 * is pinned by (1) closed-form known-answer tests derived from the SDF formulas
 * and (2) an independently written NumPy float32 restatement that must agree
 * with it (tests/test_oracle_nversion.py).
 *
 * Arithmetic is strict IEEE-754 binary32: build with -ffp-contract=off and
 * without -ffast-math.  HLSL intrinsics are expanded in ONE canonical way:
 *   dot3(a,b)     = a.x*b.x + a.y*b.y + a.z*b.z            (left to right)
 *   length(v)     = sqrt(dot(v,v))
 *   normalize(v)  = v * (1/sqrt(dot(v,v)))
 *   lerp(a,b,t)   = a + t*(b-a)
 *   clamp(x,l,h)  = min(max(x,l),h)
 *   min/max       = IEEE minNum/maxNum (a NaN operand is dropped)
 */
#ifndef RM_ORACLE_H
#define RM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* primitive kinds, in the order scene() uses them (Assets/Raymarching.cginc:351-354) */
enum { RMO_SPHERE = 0, RMO_BOX = 1, RMO_LINK = 2, RMO_CYLINDER = 3, RMO_TORUS = 4 };
/* combine operator applied when a primitive's subtree is merged upwards */
/* op bits of a primitive.  Every internal node of the pairwise tree has exactly one leaf that
 * is the leftmost leaf of its right operand; that leaf's SMAX bit chooses smax (cginc:120-123)
 * over smin (cginc:110-113) for the node.  NEGATE flips the sign of the primitive's distance
 * (the complement solid), so smax with a negated primitive is a smooth subtraction. */
enum { RMO_OP_SMIN = 0, RMO_OP_SMAX = 1, RMO_OP_NEGATE = 2 };

typedef struct {
    int32_t type;       /* RMO_* */
    int32_t op;         /* RMO_OP_* bits */
    float   pos[3];     /* _Positions row   (Assets/Raymarching.cs:61) */
    float   quat[4];    /* _Rotations row x,y,z,w (Assets/Raymarching.cs:71) */
    float   scale[3];   /* _Scales row      (Assets/Raymarching.cs:79) */
    float   color[3];   /* rgb assigned after the primitive call (cginc:356-359) */
    float   params[4];  /* sphere: r | box: b.xyz | link: le,r1,r2 | cylinder: h,r | torus: t.x,t.y */
} rmo_prim;

typedef struct {
    /* matrices are column-major float[16] exactly as a Unity Matrix4x4 is laid
     * out; row i = {m[i], m[i+4], m[i+8], m[i+12]} */
    float frustum_corners[16];   /* _FrustrumCorners   rows TL,TR,BR,BL (Raymarching.cs:50-53) */
    float cam_inv_view[16];      /* _CameraInvViewMatrix (Raymarching.cs:87) */
    float positions[16];         /* _Positions */
    float rotations[16];         /* _Rotations */
    float scales[16];            /* _Scales */
    float cam_pos[3];            /* _WorldSpaceCameraPos (Raymarching.shader:167) */
    float light[4];              /* _WorldSpaceLightPos0 (Raymarching.shader:109,199) */
    float max_steps;             /* _MaxSteps */
    float max_dist;              /* _MaxDist */
    float step_factor;           /* _StepFactor */
    float contact_threshold;     /* _ContactThreshold (initial value; overwritten by the march) */
    float normal_sample_scale;   /* _NormalSampleScale */
    float far_plane;             /* _ProjectionParams.z */
    float smooth_k;              /* k of smin; the reference hard-codes the default .5 (cginc:110) */
    float background[4];         /* constant _MainTex colour in [0,1] used when src == NULL */
    int32_t n_prims;             /* 0: the reference's fixed 4-primitive scene from the matrices */
    const rmo_prim *prims;       /* n_prims entries when n_prims > 0 */
    int32_t shade_miss_pixels;   /* 1: run normal+shadow on miss pixels too, as the shader does
                                    (their result is discarded by the select at shader:197-202) */
    int32_t scene_body;          /* RMO_BODY_*: what scene() returns */
    float fractal_scale;         /* _FractalScale (cginc:12) */
    float fractal_rotation[3];   /* _FractalRotationX/Y/Z (cginc:13) */
    float fractal_power;         /* e of mandelbulb/mandelbrot; <= 0: the function's default (7 / 2) */
    /* lines the reference keeps commented out in frag (all 0 = the shader as it stands) */
    int32_t fx_step_heat;        /* col += r.steps / 100 * _StepFactor          (shader:203) */
    int32_t fx_debug_view;       /* 1 shadow (shader:200), 2 normal*.5+.5 (:201), 3 clamp(scene(hitpoint).r*2.5-1.75,0,1) (:198) */
    int32_t fx_volumetric;       /* light integration along the ray              (shader:206-212) */
    int32_t fx_bounce;           /* one refract/reflect round (shader:178-187); 2: the refracting branch calls the reference's
                                    own refract2(rd, normal, _RefractiveIndex) (shader:142-147) instead of refract(rd, normal, 1.1) */
    /* helpers the reference defines and calls from nowhere, wired in as this build defines (0 = the shader as it stands):
     * the `light l` frag builds at shader:157-162 = {range 1000, intensity 1, position _LightPosition, color (.8,.7,.6)} */
    int32_t fx_light_model;      /* 1: getbrightness(hitpoint, normal, l) (shader:92-100), 2: getbrightnesshard (shader:84-90)
                                    takes the place of dot(normal, _WorldSpaceLightPos0)*.5+.5 in the shade (shader:199) */
    int32_t fx_scatter;          /* col += getscatter(r, l) (shader:129-140) after the shade, like the light integration */
    float light_position[3];     /* _LightPosition (shader:4,31) */
    float refractive_index;      /* _RefractiveIndex (shader:11,36) */
    /* domain operators the reference defines but never calls, applied to p at the top of scene():
     * 1 repetition p = shmod(p, period) on the axes whose period is > 0 (cginc:84-99),
     * 2 boxfold(p, size) (cginc:156-159), 3 spherefold(p, size) (cginc:161-165); 0 = none */
    int32_t domain_op;
    float domain_period[3];
    float domain_size;
    /* Shadow-map consumption.  The reference declares `sampler2D _ShadowMap; float4x4 _ShadowMapVPMatrix;`
     * (Raymarching.shader:45-46), has the binding commented out (Raymarching.cs:91-93) and no code that samples the
     * map, so the lookup is this build's definition (parity unpinned by the reference):
     *   0  the shader as it stands: getshadow() marches (shader:102-123)
     *   1  perspective map (the ShadowMap.Render pass): c = mul(_ShadowMapVPMatrix, float4(q,1)); the point is lit
     *      unless c.w > 0, uv = c.xy / c.w * .5 + .5 lies in [0,1)^2 and c.w > map[uv] * (1 + bias)
     *      (c.w is the eye depth the depth pass stores as r.length)
     *   2  directional map (the ShadowMapDirectional pass): r = q - BL; u = dot(r,e1)/dot(e1,e1), v = dot(r,e2)/dot(e2,e2)
     *      with e1 = BR - BL, e2 = TL - BL, d = dot(r, forward); lit unless uv in [0,1)^2 and d > map[uv] * (1 + bias)
     * q = hitpoint + normal * _ContactThreshold * 2, the point getshadow() starts from (shader:107); the map is
     * point-sampled at texel (floor(u*w), floor(v*h)), row 0 = bottom; the result (1 lit, 0 shadowed) takes the
     * place of getshadow()'s in frag (shader:195). */
    int32_t shadow_mode;
    const float *shadow_map;     /* shadow_w x shadow_h depths (r.length of a depth pass) */
    int32_t shadow_w, shadow_h;
    float shadow_vp[16];         /* _ShadowMapVPMatrix, column-major like every matrix here */
    float shadow_corners[12];    /* directional: TL,TR,BR,BL as ShadowMapDirectional.cs:11-16 orders them */
    float shadow_forward[3];
    float shadow_bias;
    /* _Time, Unity's built-in (t/20, t, 2t, 3t); only .y is read, by the composite body and the animated angles below */
    float time[4];
    /* 1: the three lines cginc:346-348 run: _FractalRotationX/Y/Z = sin(_Time.y)*.5, sin(_Time.y*1.414)*.5,
     * sin(_Time.y*1.618)*.5 replace the material's values (uniform-only: evaluated once per frame, sin as the double libm value
     * rounded to float like every cos/sin of a uniform angle here) */
    int32_t animate_fractal;
} rmo_uniforms;

/* scene() bodies.  PRIMS is the reference's active code (cginc:336-363).  The fractal bodies are
 * the functions the reference defines but only calls from a commented-out line (cginc:349); here
 * they get the wrapper every primitive gets, with slot 0 of the three matrices:
 *   p0 = rotate(p - _Positions[0], _Rotations[0]);  o = body(p0 / _Scales[0]);  o.w *= min(_Scales[0]) */
enum { RMO_BODY_PRIMS = 0, RMO_BODY_MENGER = 1, RMO_BODY_SIERPINSKI = 2, RMO_BODY_MANDELBULB = 3, RMO_BODY_MANDELBROT = 4,
       /* terrain(float2) (cginc:315-330) returns a height that no code of the reference turns into a distance; this
        * build's definition of the body: t = terrain(p.xz); float4(t.rgb, p.y - t.w) */
       RMO_BODY_TERRAIN = 5,
       /* the line the reference keeps commented out at cginc:349, literally -- on the world-space p, without the slot-0 wrapper:
        *   mandelbulb(rotate(p, float3(0, 0, _Time.y)),
        *              menger(rotate(p, float3(_Time.y*.25, _Time.y*1.4*.25, _Time.y*1.6*.25))).w + 1.1)
        * i.e. a mandelbulb whose exponent varies from point to point with the distance to a (differently rotated) menger sponge */
       RMO_BODY_COMPOSITE = 6 };

typedef struct {
    uint64_t n_pixels;
    uint64_t n_hit;          /* pixels with r.hit                                    */
    uint64_t sum_steps_p;    /* scene() evaluations in the primary march, all pixels */
    uint64_t sum_steps_s;    /* scene() evaluations in the shadow march, hit pixels  */
    uint64_t sum_steps_s_all;/* ... including miss pixels when shade_miss_pixels     */
} rmo_stats;

/* Renders rows [y0,y1) of a w x h frame.  Row 0 is the BOTTOM row (uv origin
 * bottom-left, cginc:30).  All output pointers address full-frame buffers and may
 * be NULL individually.  src_rgba8/scene_depth may be NULL (constant background /
 * far plane everywhere).  threads <= 0 selects one thread per online CPU. */
int rmo_render(const rmo_uniforms *u,
               const uint8_t *src_rgba8, const float *scene_depth,
               uint8_t *dst_rgba8, float *dst_depth, uint8_t *dst_hit,
               float *dst_rgb_f32,            /* 3 floats/pixel, unquantised colour */
               uint16_t *dst_steps_p, uint16_t *dst_steps_s,
               int w, int h, int y0, int y1, int threads, rmo_stats *stats);

/* depth-only passes (raymarch() only; length written, hit optional) */
int rmo_shadowmap_perspective(const rmo_uniforms *u, float *dst_depth, uint8_t *dst_hit,
                              int w, int h, int threads, rmo_stats *stats);
int rmo_shadowmap_directional(const rmo_uniforms *u, const float corners[12], const float forward[3],
                              float *dst_depth, uint8_t *dst_hit,
                              int w, int h, int threads, rmo_stats *stats);

/* single-function probes for the known-answer tests */
float rmo_sphere(const float p[3], float r);
float rmo_box(const float p[3], const float b[3]);
float rmo_link(const float p[3], float le, float r1, float r2);
float rmo_cylinder(const float p[3], float h, float r);
float rmo_torus(const float p[3], float tx, float ty);
float rmo_mod(float x, float m);
float rmo_shmod(float x, float m);
void rmo_boxfold(const float p[3], float b, float out[3]);
void rmo_spherefold(const float p[3], float R, float out[3]);
void  rmo_rotate_q(const float v[3], const float q[4], float out[3]);
float rmo_remap(float x, float o1, float o2, float n1, float n2);
void  rmo_smin4(const float a[4], const float b[4], float k, float out[4]);
float rmo_c_sin(float x);
float rmo_c_cos(float x);
float rmo_c_acos(float x);
float rmo_c_asin(float x);
float rmo_c_atan2(float y, float x);
float rmo_c_log(float x);
float rmo_c_pow(float x, float e);
float rmo_c_atan(float x);
float rmo_rand(const float co[2]);
float rmo_noise(const float st[2]);
float rmo_terrain(const float x[2]);
float rmo_getbrightness(const float p[3], const float n[3], const float light_pos[3], int hard);
float rmo_getscatter(const float ro[3], const float rd[3], float length, const float light_pos[3]);
void  rmo_refract2(const float i[3], const float n[3], float eta, float out[3]);
void  rmo_mandelbulb(const float p[3], float e, float out[4]);
void  rmo_mandelbrot(const float p[3], float e, float out[4]);
void  rmo_menger(const rmo_uniforms *u, const float p[3], float out[4]);
void  rmo_sierpinski(const rmo_uniforms *u, const float p[3], float out[4]);
void  rmo_smax4(const float a[4], const float b[4], float k, float out[4]);
float rmo_smin1(float a, float b, float k);                 /* cginc:105-108 */
float rmo_smax1(float a, float b, float k);                 /* cginc:115-118 */
float rmo_sq(float x);                                      /* cginc:68-70 */
void  rmo_sq3(const float x[3], float out[3]);              /* cginc:72-74 */
float rmo_luminance(const float c[3]);                      /* cginc:151-154 */
void  rmo_getnormal(const rmo_uniforms *u, const float p[3], float out[3], float raw[3]);   /* shader:80-82 */
void  rmo_scene(const rmo_uniforms *u, const float p[3], float out[4]);
void  rmo_pixel_ray(const rmo_uniforms *u, int x, int y, int w, int h, float rd[3]);
int   rmo_vert_corner_index(float uvx, float uvy);
/* raymarch() from an arbitrary origin/direction: returns hit, writes length and eval count */
int   rmo_raymarch(const rmo_uniforms *u, const float ro[3], const float rd[3], float depth,
                   float *length, int *evals);
int   rmo_online_cpus(void);

#ifdef __cplusplus
}
#endif
#endif
