// rm_api.cpp -- the C ABI of librmhip.so (include/rmhip.h): a Unity-Material-like
// property bag per context, and blit entry points that turn the bag into one
// kernel-argument block and launch the march.
//
// Compiled with -ffp-contract=off: the uniform-only sub-expressions evaluated
// here (corner rays, quaternion terms, min(scale)) must round as the shader's
// own fp32 evaluation would.
#include "../../include/rmhip.h"

#include <hip/hip_runtime.h>

#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rm_device.h"
#include "rm_launch.h"
#include "rm_scratch.h"

// The event behind every launch only orders work of this GPU (the next launch on another stream, the counter
// reduction); without the system-scope fence its record costs nothing measurable, with it 3 us per frame.
#define RM_EVENT_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)

namespace {

thread_local std::string g_create_error;

const char *const kDeclared[] = {
    // Assets/Raymarching.cginc:1-13
    "_FrustrumCorners", "_CameraInvViewMatrix", "_MaxSteps", "_MaxDist", "_StepFactor", "_ContactThreshold",
    "_Positions", "_Rotations", "_Scales", "_FractalScale", "_FractalRotationX", "_FractalRotationY",
    "_FractalRotationZ",
    // Assets/Raymarching.shader:3-18,30-46
    "_MainTex", "_LightPosition", "_NormalSampleScale", "_Reflections", "_RefractiveIndex", "_Color1", "_Color2",
    "_Glossiness", "_Skybox", "_CameraDepthTexture", "_ShadowMap", "_ShadowMapVPMatrix",
    // Unity built-ins the shader reads, explicit here
    "_WorldSpaceCameraPos", "_WorldSpaceLightPos0", "_ProjectionParams",
    // stands in for _MainTex when no source image is passed
    "_BackgroundColor",
    // Unity's built-in time vector (t/20, t, 2t, 3t), read by the commented lines cginc:346-349; explicit here.
    // _RmAnimateFractal 1 = the three lines cginc:346-348 (the fold angles follow sin(_Time.y))
    "_Time", "_RmAnimateFractal",
    // which scene() body runs: 0 primitives (cginc:336-363), 1 menger (:295-313), 2 sierpinski (:264-293)
    "_RmSceneBody",
    // exponent of the mandelbulb / mandelbrot bodies; 0 = the functions' defaults 7 / 2 (cginc:194,229)
    "_RmFractalPower",
    // lines of frag that the reference keeps commented out: col += r.steps/100*_StepFactor (shader:203), the
    // alternative outputs (:198-201), the light integration along the ray (:206-212), the bounce loop (:178-187)
    "_RmStepHeat", "_RmDebugView", "_RmVolumetricLight", "_RmBounce",
    // helpers the reference defines and never calls: getbrightness / getbrightnesshard (shader:84-100) in the shade,
    // getscatter (shader:129-140) added to the colour; _RmBounce 2 refracts with refract2 (shader:142-147)
    "_RmLightModel", "_RmScatter",
    // shadow-map lookup in place of the shadow march: what reads _ShadowMap / _ShadowMapVPMatrix (shader:45-46)
    "_RmShadowMode", "_RmShadowBias", "_RmShadowMapCorners", "_RmShadowMapForward",
};

typedef std::array<float, 4> vec4;
typedef std::array<float, 16> mat4;

}  // namespace

// Host-side state of the persistent kernel's tile queue together with the device buffer it describes.
// The buffer is either owned by a context (grown on demand) or a slice of the per-device static scratch
// array (fixed capacity, shared by every context on that GPU).
struct QueueState {
    uint32_t *buf = nullptr; size_t cap_bytes = 0;   // counters | heavy_count | heavy_list | bitmaps
    uint32_t bm_cap = 0;           // capacity of one heavy bitmap in words
    uint32_t epoch = 0;
    bool zeroed = false;           // the buffer holds zeros and no launch has used it since
    bool dirty = false;            // a queue launch failed: counters/lists are in an unknown rotation state
    bool history = false;          // the last launch appended tile costs for a frame of the shape below
    uint32_t last_tiles = 0, last_tx = 0, last_ty = 0; int last_mode = -1;
    // what the last single-view frame of the 4-slot scene looked at: a frame that looks elsewhere gets predicted tile costs
    // instead of that frame's (launch_queue)
    bool have_view = false;
    RmView last_view; float last_pos[4][3]; float last_light[3];
};

// Device memory a launch works in beyond the caller's images: queue state, counter slots, reduced counters.
struct WorkMem {
    QueueState *q = nullptr;
    uint32_t **partials = nullptr; size_t *partials_cap = nullptr;   // bytes
    unsigned long long *stats = nullptr;
    bool fixed = false;            // static scratch: nothing can grow, sizes beyond capacity are refused
    bool capturing = false;        // fixed only: the caller's stream is under graph capture, the launch may be replayed any number of times
};

struct rm_ctx {
    int device = 0;
    bool dev_ready = false;        // stream, event and d_stats exist
    bool props_ready = false;      // n_cus / lds_per_block are known
    std::string err;
    std::map<std::string, float> floats;
    std::map<std::string, vec4> vectors;
    std::map<std::string, mat4> matrices;
    std::vector<rm_primitive> prims;
    bool prims_dirty = false;
    int kernel = RM_KERNEL_AUTO;
    const float *shadow_map = nullptr; int shadow_w = 0, shadow_h = 0;      // _ShadowMap: caller-owned device depth plane

    hipStream_t stream = nullptr;           // used by the host-pointer entry points and by rm_get_stats
    hipEvent_t ev_last = nullptr;           // recorded behind every launch of the allocating entry points
    bool ev_recorded = false;
    unsigned long long *d_stats = nullptr;  // reduced counters of the last launch (filled by rm_get_stats)
    uint32_t *d_partials = nullptr; size_t d_partials_cap = 0;   // per-block counter slots of the last launch
    RmPrimDev *d_prims = nullptr;  size_t d_prims_cap = 0;
    RmView *d_views = nullptr;     size_t d_views_cap = 0;
    uint32_t *d_leaf_stats = nullptr; size_t d_leaf_stats_cap = 0;   // per wave of the exact-culling table kernel: leaves visited / possible
    unsigned last_leaf_waves = 0;  // waves of the last launch that wrote leaf statistics (0: it was no culling launch)
    void *d_stage[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // src, sdepth, rgba, depth, hit, steps
    size_t d_stage_cap[6] = {0, 0, 0, 0, 0, 0};
    // rm_blit_async: two staging sets, each with a stream of its own that runs uploads, march and copies back of the calls
    // that use the set, in order (set 0 is d_stage on `stream`, shared with the synchronous entry points); per set the
    // event "outputs are in the caller's buffers"
    void *d_stage_b[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t d_stage_b_cap[6] = {0, 0, 0, 0, 0, 0};
    hipStream_t stream_b = nullptr;
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    bool async_used[2] = {false, false};    // the set has a call whose events were recorded
    long long async_issued = 0;             // tickets handed out so far (ticket t uses set t & 1)

    QueueState q;                  // persistent-kernel queue state in memory this context owns
    int n_cus = 0;
    size_t lds_per_block = 0;
    std::map<uint64_t, int> occ;   // resident blocks per CU, keyed by kernel family, variant and dynamic LDS bytes

    int inv_w_for = 0, inv_h_for = 0;       // frame sizes the reciprocals below were verified for
    float inv_w = 0.0f, inv_h = 0.0f;

    // the last launch: where its counters are and how to wait for it
    uint32_t *last_partials = nullptr;
    unsigned long long *last_stats = nullptr;
    unsigned last_blocks = 0;
    hipStream_t last_stream = nullptr;      // only used to wait for a launch that left no event (the entry point that allocates nothing)
    bool last_has_event = false;
    uint64_t last_pixels = 0;
    bool have_last = false;
    bool last_fixed = false;                // the last launch left its counters in the GPU's shared static scratch ...
    uint64_t last_scratch_gen = 0;          // ... as that scratch's launch number `last_scratch_gen`
};

namespace {

int fail(rm_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define RM_HIP(c, expr)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), RM_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Makes the context's device current for one entry point and puts the caller's device back on return:
// a plugin must not leave the host thread on another GPU than it found it.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int device) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev != device) {
            e = hipSetDevice(device);
            switched = e == hipSuccess;
        }
        return e;
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

// No C++ exception may cross the C ABI (allocation failures of the property maps, strings, vectors).
#define RM_TRY try {
#define RM_CATCH(c)                                                              \
    } catch (const std::bad_alloc &) {                                           \
        return fail_noexcept((c), RM_E_NOMEM, "out of host memory");             \
    } catch (...) {                                                              \
        return fail_noexcept((c), RM_E_ARG, "unexpected C++ exception");         \
    }

int fail_noexcept(rm_ctx *c, int code, const char *msg) noexcept {
    try { if (c) c->err = msg; } catch (...) {}
    return code;
}

mat4 mat_identity() {
    mat4 m{};
    m[0] = m[5] = m[10] = m[15] = 1.0f;
    return m;
}
void set_row(mat4 &m, int i, float x, float y, float z, float w) { m[i] = x; m[i + 4] = y; m[i + 8] = z; m[i + 12] = w; }
void get_row(const mat4 &m, int i, float r[4]) { r[0] = m[i]; r[1] = m[i + 4]; r[2] = m[i + 8]; r[3] = m[i + 12]; }

// GetFrustumCorners() for a given vertical fov and aspect (Assets/Raymarching.cs:32-56)
mat4 frustum_corners(float fov_deg, float aspect) {
    const float deg2rad = 0.0174532924f;
    float half = fov_deg * 0.5f;
    float t = (float)std::tan((double)(half * deg2rad));
    float a = t * aspect;
    mat4 m = mat_identity();
    set_row(m, 0, -a, t, -1.0f, 0.0f);
    set_row(m, 1, a, t, -1.0f, 0.0f);
    set_row(m, 2, a, -t, -1.0f, 0.0f);
    set_row(m, 3, -a, -t, -1.0f, 0.0f);
    return m;
}

// Values serialized in Assets/Raymarching.mat:62-105 and the objects of
// Assets/Scenes/SampleScene.unity (camera :182-222, "Position N" transforms
// :138-140,359-361,389-391,488-490, Directional Light :458).
void load_defaults(rm_ctx *c) {
    c->floats["_ContactThreshold"] = 0.002f;
    c->floats["_FractalRotationX"] = 0.19f;
    c->floats["_FractalRotationY"] = -0.26f;
    c->floats["_FractalRotationZ"] = 0.14f;
    c->floats["_FractalScale"] = 2.0f;
    c->floats["_RmSceneBody"] = 0.0f;
    c->floats["_RmAnimateFractal"] = 0.0f;
    c->vectors["_Time"] = {0, 0, 0, 0};
    c->floats["_RmFractalPower"] = 0.0f;
    c->floats["_RmStepHeat"] = 0.0f;
    c->floats["_RmDebugView"] = 0.0f;
    c->floats["_RmVolumetricLight"] = 0.0f;
    c->floats["_RmBounce"] = 0.0f;
    c->floats["_RmLightModel"] = 0.0f;
    c->floats["_RmScatter"] = 0.0f;
    c->floats["_RmShadowMode"] = 0.0f;
    c->floats["_RmShadowBias"] = 0.01f;
    c->matrices["_ShadowMapVPMatrix"] = mat_identity();
    c->matrices["_RmShadowMapCorners"] = mat_identity();
    c->vectors["_RmShadowMapForward"] = {0, 0, 1, 0};
    c->floats["_RmDomainOp"] = 0.0f;
    c->floats["_RmDomainSize"] = 1.0f;              // default b of boxfold (cginc:156) and R of spherefold (cginc:161)
    c->vectors["_RmDomainPeriod"] = {0, 0, 0, 0};
    c->floats["_Glossiness"] = 0.2f;
    c->floats["_MaxDist"] = 50.0f;
    c->floats["_MaxSteps"] = 400.0f;
    c->floats["_NormalSampleScale"] = 0.00005f;
    c->floats["_Reflections"] = 3.12f;
    c->floats["_RefractiveIndex"] = 1.0f;
    c->floats["_StepFactor"] = 0.7f;
    c->vectors["_Color1"] = {1, 1, 1, 1};
    c->vectors["_Color2"] = {0.75f, 0.5f, 1, 1};
    c->vectors["_LightPosition"] = {0, 0, 0, 0};
    c->vectors["_WorldSpaceCameraPos"] = {2.0f, 1.0f, -11.01f, 0.0f};
    c->vectors["_WorldSpaceLightPos0"] = {0.0f, 0.5f, 0.8660254f, 0.0f};
    c->vectors["_ProjectionParams"] = {1.0f, 0.3f, 1000.0f, 1.0f / 1000.0f};
    c->vectors["_BackgroundColor"] = {0.19215687f, 0.3019608f, 0.4745098f, 1.0f};
    c->matrices["_FrustrumCorners"] = frustum_corners(80.0f, 16.0f / 9.0f);
    mat4 c2w = mat_identity();                     // TRS(pos, identity, 1) * diag(1,1,-1,1)
    set_row(c2w, 0, 1, 0, 0, 2.0f);
    set_row(c2w, 1, 0, 1, 0, 1.0f);
    set_row(c2w, 2, 0, 0, -1, -11.01f);
    c->matrices["_CameraInvViewMatrix"] = c2w;
    mat4 pos = mat_identity(), rot = mat_identity(), scl = mat_identity();
    set_row(pos, 0, -0.584f, 0.049f, 0.602f, 0);
    set_row(pos, 1, -0.74f, 0.929f, 0.0f, 0);
    set_row(pos, 2, -7.87f, 0.0f, 1.522f, 0);
    set_row(pos, 3, -3.074f, -0.087f, 0.08f, 0);
    for (int i = 0; i < 4; i++) { set_row(rot, i, 0, 0, 0, 1); set_row(scl, i, 1, 1, 1, 0); }
    c->matrices["_Positions"] = pos;
    c->matrices["_Rotations"] = rot;
    c->matrices["_Scales"] = scl;
}

float getf(rm_ctx *c, const char *n, float dflt) {
    auto it = c->floats.find(n);
    return it == c->floats.end() ? dflt : it->second;
}
vec4 getv(rm_ctx *c, const char *n) {
    auto it = c->vectors.find(n);
    return it == c->vectors.end() ? vec4{0, 0, 0, 0} : it->second;
}
mat4 getm(rm_ctx *c, const char *n) {
    auto it = c->matrices.find(n);
    return it == c->matrices.end() ? mat_identity() : it->second;
}

// may p/s be computed from RN(1/s) with one residual correction (rm_div_by in rm_math.h)?
bool rcp_safe(float s) {
    if (!std::isfinite(s) || s == 0.0f) return false;
    uint32_t b;
    std::memcpy(&b, &s, 4);
    int e = (int)((b >> 23) & 0xff) - 127;
    return e >= -60 && e <= 60 && (b & 0x7fffffu) != 0x7fffffu;
}

uint32_t unorm8(float x) {
    x = std::fmin(std::fmax(x, 0.0f), 1.0f);
    return (uint32_t)std::floor(x * 255.0f + 0.5f);
}

// evaluates the uniform-only parts of the scene() wrapper (cginc:62-66,336-354)
RmPrimDev prepare_prim(const rm_primitive &p, int *flags_out) {
    RmPrimDev d;
    std::memset(&d, 0, sizeof d);
    d.px = p.pos[0]; d.py = p.pos[1]; d.pz = p.pos[2];
    d.ux = -p.quat[0]; d.uy = -p.quat[1]; d.uz = -p.quat[2];
    float qw = p.quat[3];
    d.c = qw * qw - (d.ux * d.ux + d.uy * d.uy + d.uz * d.uz);
    d.w2 = 2.0f * qw;
    d.ms = std::fmin(p.scale[0], std::fmin(p.scale[1], p.scale[2]));
    d.type = p.type;
    d.cr = p.color[0]; d.cg = p.color[1]; d.cb = p.color[2];
    switch (p.type) {
    case RM_SPHERE: d.a0 = p.params[0] * .5f; break;
    case RM_BOX:    d.a0 = p.params[0] * .5f; d.a1 = p.params[1] * .5f; d.a2 = p.params[2] * .5f; break;
    default:        d.a0 = p.params[0]; d.a1 = p.params[1]; d.a2 = p.params[2]; break;
    }
    int flags = 0;
    if (p.quat[0] == 0.0f && p.quat[1] == 0.0f && p.quat[2] == 0.0f && p.quat[3] == 1.0f) flags |= RM_PF_IDENT_ROT;
    if (p.scale[0] == 1.0f && p.scale[1] == 1.0f && p.scale[2] == 1.0f) flags |= RM_PF_UNIT_SCALE;
    if (rcp_safe(p.scale[0]) && rcp_safe(p.scale[1]) && rcp_safe(p.scale[2])) flags |= RM_PF_RCP;
    d.rx = 1.0f / p.scale[0]; d.ry = 1.0f / p.scale[1]; d.rz = 1.0f / p.scale[2];      // exact-division use only under RM_PF_RCP
    if (p.op & (RM_OP_SMAX | RM_OP_NEGATE)) flags |= RM_PF_CSG;
    *flags_out = flags;
    d.nsgn = (p.op & RM_OP_NEGATE) ? -1.0f : 1.0f;
    d.osgn = (p.op & RM_OP_SMAX) ? -1.0f : 1.0f;
    d.sx = p.scale[0]; d.sy = p.scale[1]; d.sz = p.scale[2];
    return d;
}

void legacy_prims(rm_ctx *c, rm_primitive out[4]) {
    static const float col[4][3] = {{1, .25f, .25f}, {.25f, 1, .25f}, {.25f, .25f, 1}, {1, 1, .25f}};   // cginc:356-359
    static const float par[4][4] = {{1, 0, 0, 0}, {1, 1, 1, 0}, {5, 5, .1f, 0}, {.5f, .5f, 0, 0}};     // cginc:351-354 defaults
    mat4 P = getm(c, "_Positions"), R = getm(c, "_Rotations"), S = getm(c, "_Scales");
    for (int i = 0; i < 4; i++) {
        float p[4], q[4], s[4];
        get_row(P, i, p); get_row(R, i, q); get_row(S, i, s);
        out[i].type = i; out[i].op = 0;
        std::memcpy(out[i].pos, p, 12);
        std::memcpy(out[i].quat, q, 16);
        std::memcpy(out[i].scale, s, 12);
        std::memcpy(out[i].color, col[i], 12);
        std::memcpy(out[i].params, par[i], 16);
    }
}

// mul(_CameraInvViewMatrix, corner.xyz): float4x4 x float3 = upper-left 3x3 (cginc:31)
void corner_world(const mat4 &c2w, const mat4 &fc, int row, float out[3]) {
    float v[4];
    get_row(fc, row, v);
    for (int i = 0; i < 3; i++) {
        float r[4];
        get_row(c2w, i, r);
        out[i] = r[0] * v[0] + r[1] * v[1] + r[2] * v[2];
    }
}

void make_view(const mat4 &fc, const mat4 &c2w, const float cam_pos[3], RmView *v) {
    std::memset(v, 0, sizeof *v);
    v->ro[0] = cam_pos[0]; v->ro[1] = cam_pos[1]; v->ro[2] = cam_pos[2];
    corner_world(c2w, fc, 0, v->tl);      // rows TL,TR,BR,BL (Assets/Raymarching.cs:50-53)
    corner_world(c2w, fc, 1, v->tr);
    corner_world(c2w, fc, 2, v->br);
    corner_world(c2w, fc, 3, v->bl);
}

// what every launch needs to know about the GPU; no memory is allocated here
int ensure_props(rm_ctx *c) {
    if (c->props_ready) return RM_OK;
    int n = 0, lds = 0;
    RM_HIP(c, hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device));
    RM_HIP(c, hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device));
    c->n_cus = n > 0 ? n : 256;
    c->lds_per_block = lds > 0 ? (size_t)lds : 65536;
    c->props_ready = true;
    return RM_OK;
}

// first use of the device by an entry point that may allocate (the caller has made it current through a DeviceScope)
int ensure_device(rm_ctx *c) {
    if (c->dev_ready) return RM_OK;
    int rc = ensure_props(c);
    if (rc) return rc;
    if (!c->stream) RM_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (!c->ev_last) RM_HIP(c, hipEventCreateWithFlags(&c->ev_last, RM_EVENT_FLAGS));
    if (!c->d_stats) RM_HIP(c, hipMalloc((void **)&c->d_stats, 4 * sizeof(unsigned long long)));
    c->dev_ready = true;
    return RM_OK;
}

// Host-side wait for everything this context has launched.  Launches of the allocating entry points leave an
// event; the entry point that allocates nothing leaves only the stream it was given, which must still exist.
int wait_for_last(rm_ctx *c) {
    if (!c->have_last) return RM_OK;
    if (c->last_has_event) RM_HIP(c, hipEventSynchronize(c->ev_last));
    else RM_HIP(c, hipStreamSynchronize(c->last_stream));
    return RM_OK;
}

// Orders `stream` behind the context's previous launch, whatever stream that ran on: the queue counters,
// counter slots, primitive table and view array are shared by consecutive launches.
int order_after_last(rm_ctx *c, hipStream_t stream, bool capturing = false) {
    if (!c->have_last) return RM_OK;
    if (c->last_has_event) RM_HIP(c, hipStreamWaitEvent(stream, c->ev_last, 0));
    else if (c->last_stream != stream) {
        // no host-side wait is legal while a stream is being captured
        if (capturing) return fail(c, RM_E_ARG, "this context's previous call ran on another stream and left no event: "
                                                "synchronize that stream before capturing a graph on this one");
        RM_HIP(c, hipStreamSynchronize(c->last_stream));
    }
    return RM_OK;
}

template <class T>
int ensure_cap(rm_ctx *c, T **ptr, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return RM_OK;
    if (*ptr) { RM_HIP(c, hipFree(*ptr)); *ptr = nullptr; *cap = 0; }
    RM_HIP(c, hipMalloc((void **)ptr, bytes));
    *cap = bytes;
    return RM_OK;
}

// Fills everything of RmParams that comes from the property bag.
int build_params(rm_ctx *c, RmParams *U, hipStream_t stream) {
    std::memset(U, 0, sizeof *U);
    vec4 cam = getv(c, "_WorldSpaceCameraPos");
    make_view(getm(c, "_FrustrumCorners"), getm(c, "_CameraInvViewMatrix"), cam.data(), &U->view);
    vec4 L = getv(c, "_WorldSpaceLightPos0");
    U->light[0] = L[0]; U->light[1] = L[1]; U->light[2] = L[2];          // float4 -> float3 (shader:109)
    float max_steps = getf(c, "_MaxSteps", 100.0f);
    float max_dist = getf(c, "_MaxDist", 100.0f);
    {
        float steps = std::fmin(2000.0f, max_steps);          // (float)i < steps  <=>  i < ceil(steps) for integer i >= 0
        U->n_steps = steps > 0.0f ? (int32_t)std::ceil(steps) : 0;
    }
    U->max_dist = max_dist;
    U->shadow_depth = std::fmin(2000.0f, max_dist);
    U->far_plane = getv(c, "_ProjectionParams")[2];
    U->step_factor = getf(c, "_StepFactor", 1.0f);
    U->nss_raw = getf(c, "_NormalSampleScale", 0.01f);
    U->nss = std::fmax(0.0f, U->nss_raw);
    vec4 bg = getv(c, "_BackgroundColor");
    U->bg[0] = bg[0]; U->bg[1] = bg[1]; U->bg[2] = bg[2];
    U->bg_rgba8 = unorm8(bg[0]) | (unorm8(bg[1]) << 8) | (unorm8(bg[2]) << 16) | 0xFF000000u;
    U->band_rows = 1; U->part = 0; U->n_parts = 1;
    U->n_poses = 1;

    int flags = RM_SF_IDENT_ROT | RM_SF_UNIT_SCALE | RM_SF_RCP;
    bool lipschitz = false;
    if (c->prims.empty()) {
        rm_primitive lp[4];
        legacy_prims(c, lp);
        lipschitz = getf(c, "_RmPairSkip", 1.0f) != 0.0f;
        for (int i = 0; i < 4; i++) {
            int pf = 0;
            U->prims4[i] = prepare_prim(lp[i], &pf);
            flags &= pf;
            // a primitive's distance changes no faster than its argument moves only for finite, positive scales and
            // a quaternion that does not stretch (|q| within 1e-3 of 1: rotate() scales lengths by |q|^2)
            for (int k = 0; k < 3; k++) lipschitz = lipschitz && std::isfinite(lp[i].scale[k]) && lp[i].scale[k] > 0.0f;
            const float q2 = lp[i].quat[0] * lp[i].quat[0] + lp[i].quat[1] * lp[i].quat[1] + lp[i].quat[2] * lp[i].quat[2] + lp[i].quat[3] * lp[i].quat[3];
            lipschitz = lipschitz && q2 >= 0.999f && q2 <= 1.001f;
        }
        U->n_prims = 0;
    } else {
        size_t n = c->prims.size();
        std::vector<RmPrimDev> t(n);
        bool csg = false;
        for (size_t i = 0; i < n; i++) {
            int pf = 0;
            t[i] = prepare_prim(c->prims[i], &pf);
            flags &= pf;
            csg = csg || (pf & RM_PF_CSG);
        }
        if (csg) {
            // The kernel walks the table in groups of four and merges finished groups with a
            // binary-counter stack.  A node's op is the SMAX bit of the leftmost primitive of its
            // right operand: inside a group those records are at hand; for the stack merges the bits
            // are gathered here, into record 4g for the cascade after group g and into tail_ops for
            // the final low-to-high merge of the leftovers.
            auto smax_at = [&](size_t prim) { return (c->prims[prim].op & RM_OP_SMAX) ? 1u : 0u; };
            const size_t groups = n >> 2, rem = n & 3;
            for (size_t g = 0; g < groups; g++) {
                uint32_t bits = 0;
                for (unsigned l = 0; l < 9 && ((g >> l) & 1u); l++)          // trailing ones of g
                    bits |= smax_at(4 * (g - ((size_t)1 << l) + 1)) << l;     // v covers groups g-2^l+1 .. g
                t[4 * g].type |= (int32_t)(bits << RM_CASCADE_SHIFT);
            }
            uint32_t tail = 0;
            bool have = rem != 0;
            size_t acc_first = 4 * groups;
            for (unsigned l = 0; l < 9; l++) {
                if (!(groups & ((size_t)1 << l))) continue;
                if (have) tail |= smax_at(acc_first) << l;
                acc_first = 4 * (groups & ~(((size_t)2 << l) - 1));
                have = true;
            }
            U->tail_ops = tail;
            flags |= RM_SF_CSG;
        }
        if (c->prims_dirty || !c->d_prims) {
            const size_t bytes = n * sizeof(RmPrimDev);
            int rc = ensure_cap(c, (uint8_t **)&c->d_prims, &c->d_prims_cap, bytes);
            if (rc) return rc;
            // the table must be in place before any later launch reads it, and the
            // host vectors die at return: blocking copies, done only when the scene changed,
            // after every earlier launch (on whatever stream) has finished with the old table
            if ((rc = wait_for_last(c))) return rc;
            RM_HIP(c, hipMemcpy(c->d_prims, t.data(), n * sizeof(RmPrimDev), hipMemcpyHostToDevice));
            c->prims_dirty = false;
        }
        U->prim_table = c->d_prims;
        U->n_prims = (int)n;
        U->lds_prims = (n <= RM_LDS_PRIMS && getf(c, "_RmLdsTable", 1.0f) != 0.0f) ? (int)n : 0;
    }
    const int body = (int)getf(c, "_RmSceneBody", 0.0f);
    if (body != RM_BODY_PRIMS) {
        if (body < RM_BODY_MENGER || body > RM_BODY_COMPOSITE)
            return fail(c, RM_E_ARG, "_RmSceneBody must be 0 (primitives), 1 (menger), 2 (sierpinski), 3 (mandelbulb), 4 (mandelbrot), "
                                     "5 (terrain) or 6 (the composite of cginc:349)");
        // the wrapper transform is slot 0 of the matrices, whatever table is set
        rm_primitive lp[4];
        legacy_prims(c, lp);
        int pf = 0;
        U->prims4[0] = prepare_prim(lp[0], &pf);
        flags = pf & (RM_SF_IDENT_ROT | RM_SF_UNIT_SCALE | RM_SF_RCP);
        U->n_prims = 0; U->lds_prims = 0;
        U->body = body;
        const float scale = getf(c, "_FractalScale", 2.0f);
        U->f_scale = scale;
        U->f_scale_m1 = scale - 1.0f;
        const char *names[3] = {"_FractalRotationX", "_FractalRotationY", "_FractalRotationZ"};
        const float t = getv(c, "_Time")[1];
        float ang[3];
        for (int i = 0; i < 3; i++) ang[i] = getf(c, names[i], 0.0f);
        if (getf(c, "_RmAnimateFractal", 0.0f) != 0.0f) {               // cginc:346-348; uniform-only, so taken once per frame
            ang[0] = (float)std::sin((double)t) * .5f;
            ang[1] = (float)std::sin((double)(t * 1.414f)) * .5f;
            ang[2] = (float)std::sin((double)(t * 1.618f)) * .5f;
        }
        auto cos_sin = [](const float a[3], float out[6]) {
            for (int i = 0; i < 3; i++) { out[2 * i] = (float)std::cos((double)a[i]); out[2 * i + 1] = (float)std::sin((double)a[i]); }
        };
        cos_sin(ang, U->f_rot);
        const float a_time[3] = {0.0f, 0.0f, t};                       // cginc:349
        const float a_sponge[3] = {t * .25f, t * 1.4f * .25f, t * 1.6f * .25f};
        cos_sin(a_time, U->f_rot_time);
        cos_sin(a_sponge, U->f_rot_sponge);
        const float power = getf(c, "_RmFractalPower", 0.0f);
        U->f_power = power > 0.0f ? power : (body == RM_BODY_MANDELBULB ? 7.0f : 2.0f);
        U->f_pow3 = (float)std::pow(3.0, -12.0);
        for (int i = 0; i < 16; i++) U->f_pows[i] = (float)std::pow((double)scale, -(double)i);
    }
    U->scene_flags = flags | (lipschitz ? RM_SF_LIPSCHITZ : 0);
    U->fx_flags = (getf(c, "_RmStepHeat", 0.0f) != 0.0f ? RM_FX_STEP_HEAT : 0u) |
                  (getf(c, "_RmVolumetricLight", 0.0f) != 0.0f ? RM_FX_VOLUMETRIC : 0u) |
                  (getf(c, "_RmBounce", 0.0f) != 0.0f ? RM_FX_BOUNCE : 0u) |
                  ((int)getf(c, "_RmBounce", 0.0f) == 2 ? RM_FX_REFRACT2 : 0u) |
                  (getf(c, "_RmScatter", 0.0f) != 0.0f ? RM_FX_SCATTER : 0u);
    U->light_model = (int32_t)getf(c, "_RmLightModel", 0.0f);
    if (U->light_model < 0 || U->light_model > 2) return fail(c, RM_E_ARG, "_RmLightModel must be 0, 1 (getbrightness) or 2 (getbrightnesshard)");
    {
        const vec4 lp = getv(c, "_LightPosition");
        U->light_pos[0] = lp[0]; U->light_pos[1] = lp[1]; U->light_pos[2] = lp[2];
    }
    U->refr_index = getf(c, "_RefractiveIndex", 1.0f);
    U->fx_view = (int32_t)getf(c, "_RmDebugView", 0.0f);
    if (U->fx_view < 0 || U->fx_view > 3) return fail(c, RM_E_ARG, "_RmDebugView must be 0..3");
    U->shadow_mode = (int32_t)getf(c, "_RmShadowMode", 0.0f);
    if (U->shadow_mode < 0 || U->shadow_mode > 2) return fail(c, RM_E_ARG, "_RmShadowMode must be 0 (march), 1 (perspective map) or 2 (directional map)");
    if (U->shadow_mode != 0) {
        if (!c->shadow_map) return fail(c, RM_E_ARG, "_RmShadowMode is set but no _ShadowMap is bound (rm_set_texture_device)");
        U->shadow_map = c->shadow_map; U->sm_w = c->shadow_w; U->sm_h = c->shadow_h;
        const mat4 vp = getm(c, "_ShadowMapVPMatrix");
        get_row(vp, 0, U->sm_row0); get_row(vp, 1, U->sm_row1); get_row(vp, 3, U->sm_row3);
        const mat4 k = getm(c, "_RmShadowMapCorners");            // rows TL, TR, BR, BL (Assets/ShadowMapDirectional.cs:11-16)
        float tl[4], br[4], bl[4];
        get_row(k, 0, tl); get_row(k, 2, br); get_row(k, 3, bl);
        for (int i = 0; i < 3; i++) { U->sm_bl[i] = bl[i]; U->sm_e1[i] = br[i] - bl[i]; U->sm_e2[i] = tl[i] - bl[i]; }
        U->sm_n1 = U->sm_e1[0] * U->sm_e1[0] + U->sm_e1[1] * U->sm_e1[1] + U->sm_e1[2] * U->sm_e1[2];
        U->sm_n2 = U->sm_e2[0] * U->sm_e2[0] + U->sm_e2[1] * U->sm_e2[1] + U->sm_e2[2] * U->sm_e2[2];
        const vec4 f = getv(c, "_RmShadowMapForward");
        U->sm_fwd[0] = f[0]; U->sm_fwd[1] = f[1]; U->sm_fwd[2] = f[2];
        U->sm_onebias = 1.0f + getf(c, "_RmShadowBias", 0.01f);
    }
    U->dom_op = (int32_t)getf(c, "_RmDomainOp", 0.0f);
    if (U->dom_op < RM_DOM_NONE || U->dom_op > RM_DOM_SPHEREFOLD) return fail(c, RM_E_ARG, "_RmDomainOp must be 0..3");
    const vec4 period = getv(c, "_RmDomainPeriod");
    for (int i = 0; i < 3; i++) U->dom_period[i] = period[i];
    U->dom_size = getf(c, "_RmDomainSize", 1.0f);
    return RM_OK;
}

// RN(1/n) if q0 = x*r, q = fma(fma(-q0, n, x), r, q0) equals x / n for every pixel centre x = i + 0.5,
// i < n (checked here, once per size, with the same fused operations the kernel uses); else 0
float exact_reciprocal_for_centres(int n) {
    const float fn = (float)n, r = 1.0f / fn;
    for (int i = 0; i < n; i++) {
        const float x = (float)i + 0.5f;
        const float q0 = x * r;
        const float q = std::fma(std::fma(-q0, fn, x), r, q0);
        if (q != x / fn) return 0.0f;
    }
    return r;
}

void frame_reciprocals(rm_ctx *c, int w, int h, RmParams *U) {
    if (c->inv_w_for != w) { c->inv_w = exact_reciprocal_for_centres(w); c->inv_w_for = w; }
    if (c->inv_h_for != h) { c->inv_h = exact_reciprocal_for_centres(h); c->inv_h_for = h; }
    const bool ok = c->inv_w != 0.0f && c->inv_h != 0.0f;
    U->inv_w = ok ? c->inv_w : 0.0f;
    U->inv_h = ok ? c->inv_h : 0.0f;
}

int check_size(rm_ctx *c, int w, int h) {
    if (w <= 0 || h <= 0) return fail(c, RM_E_ARG, "width and height must be positive");
    if (w > 32768 || h > 32768) return fail(c, RM_E_RANGE, "frame larger than 32768 x 32768");
    return RM_OK;
}

constexpr size_t RM_Q_HDR = 2 * RM_Q_SHARDS * RM_Q_STRIDE + 64;   // counters, then the three heavy counts on a line of their own
constexpr uint32_t RM_STATIC_TILES_MAX = 6;      // static tiles per wave at most (see launch_queue)
constexpr uint32_t RM_HEAVY_CAP = 8192;          // heavy-tile list entries kept per frame
constexpr uint32_t RM_QUEUE_MAX_TILES = 160u * 1024u;   // 20 KB of LDS bitmap per block at most (4K frame = 129600 tiles)

int ensure_partials(rm_ctx *c, WorkMem &M, unsigned slots) {
    size_t bytes = (size_t)slots * 4 * rm_slot_words();
    if (bytes <= *M.partials_cap) return RM_OK;
    if (M.fixed) return fail(c, RM_E_RANGE, "frame needs more counter slots than the static scratch holds");
    // growing the slot buffer frees the old one: earlier launches must have finished with it
    int rc = wait_for_last(c);
    if (rc) return rc;
    return ensure_cap(c, M.partials, M.partials_cap, bytes);
}

constexpr size_t queue_words(uint32_t bm_cap) { return RM_Q_HDR + 3 * (size_t)RM_HEAVY_CAP + 3 * (size_t)bm_cap; }

// RM_OK, an error, or RM_FALLBACK_TILE when the persistent kernel cannot run this launch (its dynamic LDS does
// not fit a workgroup) and the caller did not insist on it
constexpr int RM_FALLBACK_TILE = 1;

// Image coordinates (u, v in [0,1] inside the frame) of world point p under view V: a * (p - ro) = bl + u*(br - bl) + v*(tl - bl).
// false for a point behind the camera or a degenerate view.
bool project_to_view(const RmView &V, const float p[3], float &u, float &v) {
    const double ex[3] = {(double)V.br[0] - V.bl[0], (double)V.br[1] - V.bl[1], (double)V.br[2] - V.bl[2]};
    const double ey[3] = {(double)V.tl[0] - V.bl[0], (double)V.tl[1] - V.bl[1], (double)V.tl[2] - V.bl[2]};
    const double d[3] = {(double)p[0] - V.ro[0], (double)p[1] - V.ro[1], (double)p[2] - V.ro[2]};
    const double b[3] = {V.bl[0], V.bl[1], V.bl[2]};
    auto det3 = [](const double *x, const double *y, const double *z) {
        return x[0] * (y[1] * z[2] - y[2] * z[1]) - x[1] * (y[0] * z[2] - y[2] * z[0]) + x[2] * (y[0] * z[1] - y[1] * z[0]);
    };
    // u*ex + v*ey + a*(-d) = -bl
    const double nd[3] = {-d[0], -d[1], -d[2]}, nb[3] = {-b[0], -b[1], -b[2]};
    const double D = det3(ex, ey, nd);
    if (!(std::fabs(D) > 1e-30)) return false;
    const double uu = det3(nb, ey, nd) / D, vv = det3(ex, nb, nd) / D, a = det3(ex, ey, nb) / D;
    if (!(a > 0.0) || !std::isfinite(uu) || !std::isfinite(vv)) return false;
    u = (float)uu; v = (float)vv;
    return true;
}

// Has the picture moved by more than a few tiles since the frame whose tile costs the queue state holds?  The four primitives'
// positions are projected with both views; a light that turned counts too (the long tiles are shadow marches).  Only orders
// the queue: a wrong answer costs time, never a pixel.
bool looks_elsewhere(const QueueState &S, const RmParams &P) {
    const float max_shift_px = 4.0f * (float)RM_TW;
    for (int i = 0; i < 4; i++) {
        const float pn[3] = {P.prims4[i].px, P.prims4[i].py, P.prims4[i].pz};
        float u0, v0, u1, v1;
        const bool ok0 = project_to_view(S.last_view, S.last_pos[i], u0, v0), ok1 = project_to_view(P.view, pn, u1, v1);
        if (ok0 != ok1) return true;
        if (ok0 && (std::fabs(u1 - u0) * (float)P.w > max_shift_px || std::fabs(v1 - v0) * (float)P.h > max_shift_px)) return true;
    }
    double dl = 0.0, l2 = 0.0;
    for (int k = 0; k < 3; k++) { const double d = (double)P.light[k] - S.last_light[k]; dl += d * d; l2 += (double)P.light[k] * P.light[k]; }
    return !(dl <= 0.0025 * l2);          // more than 5 % of its length (NaN lights included)
}

int launch_queue(rm_ctx *c, RmParams &P, hipStream_t stream, int family, WorkMem &M, bool forced) {
    const bool streamk = family == RM_KERNEL_STREAM, fast = family == RM_KERNEL_FAST;
    QueueState &S = *M.q;
    const uint32_t tx = (uint32_t)(P.w + RM_TW - 1) / RM_TW, ty = (uint32_t)(P.rows + RM_TH - 1) / RM_TH;
    const uint64_t n_tiles64 = (uint64_t)tx * ty * (uint64_t)P.n_poses;
    if (n_tiles64 > 0x7fffffffull) return fail(c, RM_E_RANGE, "more than 2^31 tiles in one launch: split the pose batch");
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    // the heavy bitmap lives in LDS: keep history only while it fits comfortably beside 4+ blocks per CU
    // A launch recorded into a graph is replayed with the same arguments any number of times, at moments the host does
    // not see: it must neither depend on nor advance the frame-to-frame rotation of the counter and heavy-tile sets.
    // Such a launch is self-contained: its claim counters are cleared in the stream ahead of it (a memset node of the
    // same graph), it reads no history and appends none, and its kernel leaves every set alone.
    const bool solo = M.fixed && M.capturing;
    bool keep_history = !solo && !streamk && n_tiles <= RM_QUEUE_MAX_TILES;
    uint32_t bm_words = keep_history ? (n_tiles + 31) / 32 : 0;
    if (M.fixed && bm_words > S.bm_cap) { keep_history = false; bm_words = 0; }
    RmQueue &Q = P.queue;
    Q.bitmap_words = bm_words;

    // dynamic LDS of this launch: bitmap (+ staged primitive table), known before anything is asked of the runtime
    const size_t table_bytes = (size_t)P.lds_prims * sizeof(RmPrimDev);
    const size_t lds_bytes = streamk ? table_bytes : (size_t)((bm_words + 3u) & ~3u) * 4 + (P.n_prims > 0 ? table_bytes : 0);
    if (lds_bytes + 256 > c->lds_per_block) {
        if (!forced) return RM_FALLBACK_TILE;
        return fail(c, RM_E_RANGE, "the persistent kernel's tile bitmap and primitive table do not fit one workgroup's LDS for this frame");
    }
    const int vi = rm_variant_index(P), tab = P.n_prims > 0 ? 1 : 0;
    const uint64_t occ_key = ((uint64_t)(streamk ? 1 : fast ? 2 : 3) << 60) | ((uint64_t)rm_queue_shape(P) << 56) | ((uint64_t)vi << 52) | ((uint64_t)tab << 48) | (uint64_t)lds_bytes;
    auto occ_it = c->occ.find(occ_key);
    if (occ_it == c->occ.end()) {
        int o[2] = {0, 0};
        RM_HIP(c, streamk ? rm_launch_stream(P, 0, stream, o) : fast ? rm_launch_fast(P, 0, stream, o) : rm_launch_queue(P, 0, stream, o));
        occ_it = c->occ.emplace(occ_key, o[tab] > 0 ? o[tab] : 1).first;
    }

    if (!S.buf || bm_words > S.bm_cap) {
        // (never reached with static scratch: its buffer exists and bm_words was capped above)
        int rc = wait_for_last(c);
        if (rc) return rc;
        uint32_t cap = bm_words > S.bm_cap ? bm_words : S.bm_cap;
        if (cap < 64) cap = 64;
        rc = ensure_cap(c, &S.buf, &S.cap_bytes, queue_words(cap) * 4);
        if (rc) return rc;
        S.bm_cap = cap;
        S.zeroed = false; S.dirty = true;            // fresh memory: cleared below, in stream order
        S.history = false;
    }
    Q.counters = S.buf;
    Q.heavy_count = S.buf + 2 * RM_Q_SHARDS * RM_Q_STRIDE;
    Q.heavy_list = S.buf + RM_Q_HDR;
    Q.bitmaps = Q.heavy_list + 3 * (size_t)RM_HEAVY_CAP;
    Q.n_tiles = n_tiles; Q.tiles_x = tx; Q.tiles_y = ty;
    Q.tx_magic = ((1ull << 37) + tx - 1) / tx;            // tiles per pose < 2^24, tx <= 4096: (t * magic) >> 37 == t / tx
    const bool same_shape = S.history && S.last_mode == P.mode && S.last_tiles == n_tiles && S.last_tx == tx && S.last_ty == ty;
    if (solo) {
        RM_HIP(c, rm_zero_words(S.buf, 2 * RM_Q_SHARDS * RM_Q_STRIDE, stream));
    } else if (S.dirty || (keep_history && !same_shape && !S.zeroed)) {
        // The three bitmap sets were laid out for another tile count, or a launch that kept no history has
        // rotated past them without clearing: start clean.  The stream is already ordered behind the
        // context's previous launch, so the clear needs no host-side wait.
        RM_HIP(c, hipMemsetAsync(S.buf, 0, queue_words(S.bm_cap) * 4, stream));
        S.dirty = false;
        S.history = false;
    } else if (M.fixed) {
        // the shared scratch may have been used by a replayed graph since this rotation last saw it: the claim
        // counters are cleared ahead of every launch (the heavy-tile sets are never touched by a replay)
        RM_HIP(c, rm_zero_words(S.buf, 2 * RM_Q_SHARDS * RM_Q_STRIDE, stream));
    }
    {
        static const uint32_t primes[] = {7919u, 7907u, 7901u, 7883u, 7879u, 7877u, 7873u, 7867u};
        Q.scatter_mul = 1;
        if (getf(c, "_RmScatterTiles", 1.0f) != 0.0f)
            for (uint32_t pr : primes) if (n_tiles % pr != 0) { Q.scatter_mul = pr; break; }
    }
    Q.heavy_cap = RM_HEAVY_CAP;
    float thr = getf(c, "_RmHeavyTileThreshold", 24.0f);
    Q.heavy_threshold = keep_history && thr > 0.0f ? (uint32_t)thr : 0xffffffffu;
    if (solo) {
        Q.use_history = 0u;
        Q.cnt_cur = 0u; Q.cnt_clr = RM_Q_SOLO;
        Q.hv_cur = 0u; Q.hv_nxt = 1u; Q.hv_clr = 2u;       // unused: no history is read, appended or cleared
    } else {
        const uint32_t e = ++S.epoch;
        Q.use_history = (keep_history && getf(c, "_RmTileFeedback", 1.0f) != 0.0f && same_shape) ? 1u : 0u;
        Q.cnt_cur = e & 1u; Q.cnt_clr = (e + 1u) & 1u;
        Q.hv_cur = e % 3u; Q.hv_nxt = (e + 1u) % 3u; Q.hv_clr = (e + 2u) % 3u;
        S.zeroed = false;
        S.history = keep_history;
        S.last_mode = P.mode; S.last_tiles = n_tiles; S.last_tx = tx; S.last_ty = ty;
    }

    // No previous frame to order the queue by -- a context's first frame, another frame size -- or one that looked elsewhere (a
    // camera cut, a primitive or the light moved far): one ray per tile, marched ahead of the frame, fills the heavy-tile set
    // the frame kernel is about to read (rm_k_predict.hip).  Without history the three sets are all zero at this point (the
    // clear above ran, or the buffer has not been used since it was zeroed); with a stale one only the set's count has to be
    // cleared, the prediction writes every word of its bitmap.
    {
        const bool one_view = !P.views && P.n_poses == 1 && P.n_parts == 1;
        const bool may_predict = !solo && !streamk && keep_history && P.n_prims == 0 && P.body == RM_BODY_PRIMS && P.mode == RM_MODE_FRAG &&
                                 Q.heavy_threshold != 0xffffffffu && getf(c, "_RmTileFeedback", 1.0f) != 0.0f &&
                                 getf(c, "_RmPredictTiles", 1.0f) != 0.0f;
        const bool stale = may_predict && same_shape && one_view && S.have_view && looks_elsewhere(S, P);
        if (may_predict && (!same_shape || stale)) {
            // a centre ray at two thirds of the frame kernel's threshold (16 of 24) finds the long tiles without listing the cheap ones
            const float pt = getf(c, "_RmPredictThreshold", 0.0f);
            const unsigned p_thr = pt >= 1.0f ? (unsigned)pt : (2u * Q.heavy_threshold + 2u) / 3u;
            hipError_t pe = stale ? rm_zero_words(Q.heavy_count + Q.hv_cur, 1, stream) : hipSuccess;
            if (pe == hipSuccess) pe = rm_launch_predict(P, p_thr, stream);
            if (pe != hipSuccess) { S.dirty = true; S.history = false; return fail(c, RM_E_DEVICE, std::string("tile prediction launch: ") + hipGetErrorString(pe)); }
            Q.use_history = 1u;
            c->floats["_RmPredictedFrames"] = getf(c, "_RmPredictedFrames", 0.0f) + 1.0f;     // read-only diagnostic: frames ordered by a prediction so far
        }
        if (!solo) {
            S.have_view = one_view && P.n_prims == 0 && P.body == RM_BODY_PRIMS;
            if (S.have_view) {
                S.last_view = P.view;
                for (int i = 0; i < 4; i++) { S.last_pos[i][0] = P.prims4[i].px; S.last_pos[i][1] = P.prims4[i].py; S.last_pos[i][2] = P.prims4[i].pz; }
                for (int k = 0; k < 3; k++) S.last_light[k] = P.light[k];
            }
        }
    }

    unsigned per_cu = (unsigned)occ_it->second;
    const float cap_per_cu = getf(c, "_RmBlocksPerCU", 0.0f);      // 0: as many as fit
    if (cap_per_cu >= 1.0f && (unsigned)cap_per_cu < per_cu) per_cu = (unsigned)cap_per_cu;
    unsigned resident = (unsigned)c->n_cus * per_cu;
    // a workgroup renders four tiles at a time (one per wave), or one where its waves share a tile's rays
    unsigned want = (streamk && rm_stream_is_coop(P)) ? n_tiles : (n_tiles + 3) / 4;
    unsigned blocks = want < resident ? want : resident;
    // static share: a fraction of the tiles, dealt evenly over the waves that will not start on a
    // heavy tile (at most heavy_cap of them do; the kernel uses the exact count)
    // Default (property unset or negative): three quarters of a wave's fair share but at most
    // RM_STATIC_TILES_MAX tiles -- the static part only has to keep a wave busy until its first claim
    // has come back; a larger fixed share leaves big frames unbalanced (4K: 264 us with 6 tiles per
    // wave, 285 us with 24), a smaller one makes small frames wait on the counters.
    float frac = getf(c, "_RmStaticFraction", -1.0f);
    const float per_wave = (float)n_tiles / (float)(blocks * 4);
    if (frac < 0.0f) {
        const float share = 0.75f * per_wave;
        Q.static_per_wave = (uint32_t)(share < (float)RM_STATIC_TILES_MAX ? share : (float)RM_STATIC_TILES_MAX);
    } else {
        frac = frac > 1.0f ? 1.0f : frac;
        Q.static_per_wave = (uint32_t)(frac * per_wave);
    }
    int rc = ensure_partials(c, M, blocks * 4);
    if (rc) { if (!solo) S.dirty = true; return rc; }
    P.partials = *M.partials;
    {
        // every frame clears the counter set and lists of a later frame: if this launch does not happen,
        // that rotation is broken and the next launch has to start from a zeroed state
        hipError_t le = streamk ? rm_launch_stream(P, blocks, stream, nullptr)
                      : fast ? rm_launch_fast(P, blocks, stream, nullptr) : rm_launch_queue(P, blocks, stream, nullptr);
        if (le != hipSuccess) {
            if (!solo) { S.dirty = true; S.history = false; }
            return fail(c, RM_E_DEVICE, std::string("queue kernel launch: ") + hipGetErrorString(le));
        }
    }
    c->last_blocks = blocks * 4;
    return RM_OK;
}

WorkMem own_mem(rm_ctx *c) {
    WorkMem M;
    M.q = &c->q; M.partials = &c->d_partials; M.partials_cap = &c->d_partials_cap; M.stats = c->d_stats; M.fixed = false;
    return M;
}

// The static scratch array of one GPU carved into the same three pieces a context allocates for itself.  The
// queue's host-side rotation state lives here, with the memory it describes: every context that uses the
// entry point which allocates nothing on this GPU shares both.
struct DeviceScratch {
    bool ready = false;
    QueueState q;
    uint32_t *partials = nullptr; size_t partials_cap = 0;
    unsigned long long *stats = nullptr;
    uint64_t generation = 0;       // launches that have used this GPU's scratch so far: whose counters the slots hold
};
std::mutex g_scratch_mu;
DeviceScratch g_scratch[RM_SCRATCH_MAX_DEVICES];

// the context's device must be current
int device_scratch(rm_ctx *c, DeviceScratch **out) {
    if (c->device >= RM_SCRATCH_MAX_DEVICES) return fail(c, RM_E_RANGE, "device index beyond the static scratch table");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    DeviceScratch &D = g_scratch[c->device];
    if (!D.ready) {
        void *base = nullptr;
        RM_HIP(c, rm_scratch_address(c->device, &base));
        unsigned char *b = (unsigned char *)base;
        D.q.buf = (uint32_t *)b;
        D.q.cap_bytes = RM_SCRATCH_QUEUE_BYTES;
        D.q.bm_cap = RM_SCRATCH_BM_WORDS;
        D.q.dirty = true;                        // cleared, in stream order, by the first launch
        D.partials = (uint32_t *)(b + RM_SCRATCH_QUEUE_BYTES);
        D.partials_cap = RM_SCRATCH_SLOT_BYTES;
        D.stats = (unsigned long long *)(b + RM_SCRATCH_QUEUE_BYTES + RM_SCRATCH_SLOT_BYTES);
        D.ready = true;
    }
    *out = &D;
    return RM_OK;
}

// The counters of a launch of the entry point that allocates nothing stay in the GPU's shared scratch only until the
// next such launch of ANY context on that GPU.
int scratch_still_ours(rm_ctx *c) {
    if (!c->have_last || !c->last_fixed) return RM_OK;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (c->device < RM_SCRATCH_MAX_DEVICES && g_scratch[c->device].generation != c->last_scratch_gen)
        return fail(c, RM_E_ARG, "the static scratch of this GPU has been reused by a later rm_blit_device_noalloc call "
                                 "(of this or another context): the counters of this context's last call are gone");
    return RM_OK;
}

// `record`: leave an event behind the launch (every entry point but the one that allocates nothing)
int launch(rm_ctx *c, const RmParams &U, hipStream_t stream, WorkMem M, bool record) {
    RmParams P = U;
    frame_reciprocals(c, U.w, U.h, &P);
    // auto: the persistent queue kernel pays on the 4-slot scene, whose frames are a mass of cheap
    // tiles plus a few long ones; N-primitive scenes are uniformly expensive per tile and run
    // faster on the plain one-block-per-tile kernel (more registers free for the table walk)
    // (the fractal bodies are uniformly expensive too: measured 1.18 ms tile vs 1.57 ms queue at 1080p)
    // (the depth passes have no shadow march and hence no long tiles either: 4K directional pass 276 us tile vs
    // 311 us queue; keeping them off the persistent kernel also keeps their tile costs out of the frag pass's history)
    const bool forced = c->kernel != RM_KERNEL_AUTO;
    int kernel = forced ? c->kernel
               : ((U.n_prims > 0 || U.body != RM_BODY_PRIMS || U.mode != RM_MODE_FRAG) ? RM_KERNEL_TILE : RM_KERNEL_QUEUE);
    if (kernel == RM_KERNEL_COOP) kernel = RM_KERNEL_TILE;       // a tile kernel: the cooperative one where it applies (below)
    // the tolerance kernel exists for the reference's 4-slot scene only; anything else takes the automatic (exact) choice
    if (kernel == RM_KERNEL_FAST && (U.n_prims > 0 || U.body != RM_BODY_PRIMS)) {
        kernel = (U.n_prims > 0 || U.body != RM_BODY_PRIMS || U.mode != RM_MODE_FRAG) ? RM_KERNEL_TILE : RM_KERNEL_QUEUE;
    }
    if (U.body != RM_BODY_PRIMS && kernel == RM_KERNEL_STREAM) kernel = RM_KERNEL_QUEUE;      // no stream kernel for the fractal bodies
    if (U.mode == RM_MODE_FRAG && (U.fx_flags != 0 || U.fx_view != 0 || U.light_model != 0)) kernel = RM_KERNEL_TILE;   // the optional frag lines exist in the tile kernels only
    if (U.dom_op != RM_DOM_NONE) kernel = RM_KERNEL_TILE;                                          // and so do the domain operators
    // the shadow-map lookup: the persistent kernel has it for the plain frame (4-slot scene, whole frame, no source image or
    // depth texture), everything else takes the tile kernels with the optional lines
    if (U.mode == RM_MODE_FRAG && U.shadow_mode != 0 && !(kernel == RM_KERNEL_QUEUE && rm_queue_shape(P) == 4)) kernel = RM_KERNEL_TILE;
    if (U.dst_steps) kernel = RM_KERNEL_TILE;                                                      // and the per-pixel step-count plane
    int rc = order_after_last(c, stream, M.capturing);
    if (rc) return rc;
    if (kernel != RM_KERNEL_TILE) {
        c->last_leaf_waves = 0;
        rc = launch_queue(c, P, stream, kernel, M, forced);
        if (rc == RM_FALLBACK_TILE) kernel = RM_KERNEL_TILE;
        else if (rc) return rc;
    }
    if (kernel == RM_KERNEL_TILE) {
        const bool plain = U.fx_flags == 0 && U.fx_view == 0 && U.light_model == 0 && U.dom_op == RM_DOM_NONE && !U.dst_steps &&
                           !(U.mode == RM_MODE_FRAG && U.shadow_mode != 0);
        // tables: the four waves of a block share one 8x8 tile and split the table (same bits, waves that finish
        // together, a four times shorter critical chain), with or without frag's optional lines; rm_set_kernel(1) keeps
        // the one-wave-per-tile kernels
        const bool coop = c->kernel != RM_KERNEL_TILE && U.body == RM_BODY_PRIMS &&
                          U.n_prims >= RM_COOP_MIN_PRIMS && U.lds_prims == U.n_prims;
        unsigned n_blocks = coop ? rm_coop_blocks(U.w, U.rows, U.n_poses) : rm_tile_blocks(U.w, U.rows, U.n_poses);
        rc = ensure_partials(c, M, n_blocks);
        if (rc) return rc;
        P.partials = *M.partials;
        // tables of 33 to 512 primitives (from 65 with op bits): the shared tile with exact culling in the marches (same bits;
        // _RmTableSkip 0 keeps the full table walk)
        const bool skip = coop && plain && rm_skip_applies(P) && getf(c, "_RmTableSkip", 1.0f) != 0.0f;
        c->last_leaf_waves = 0;
        if (skip && !M.fixed) {
            // what the culling saved, for rm_get_leaf_stats: two words per wave
            const size_t bytes = (size_t)n_blocks * 4 * 2 * sizeof(uint32_t);
            if (bytes > c->d_leaf_stats_cap && (rc = wait_for_last(c))) return rc;
            if ((rc = ensure_cap(c, &c->d_leaf_stats, &c->d_leaf_stats_cap, bytes))) return rc;
            P.leaf_stats = c->d_leaf_stats;
            c->last_leaf_waves = n_blocks * 4;
        }
        if (skip) RM_HIP(c, rm_launch_coop_skip_tiles(P, stream));
        else if (coop) RM_HIP(c, rm_launch_coop_tiles(P, stream));
        else RM_HIP(c, rm_launch_tiles(P, stream));
        c->last_blocks = n_blocks;
    }
    c->last_partials = *M.partials;
    c->last_stats = M.stats;
    c->last_stream = stream;
    c->last_has_event = false;
    c->last_pixels = (uint64_t)U.w * (uint64_t)U.rows * (uint64_t)U.n_poses;
    c->have_last = true;
    c->last_fixed = M.fixed;
    if (record) {
        RM_HIP(c, hipEventRecord(c->ev_last, stream));
        c->last_has_event = true;
    }
    return RM_OK;
}

int stage(rm_ctx *c, int slot, size_t bytes) { return ensure_cap(c, &c->d_stage[slot], &c->d_stage_cap[slot], bytes); }

}  // namespace

extern "C" {

const char *rm_version(void) { return "rmhip 0.1 (gfx950)"; }

void UnityPluginLoad(void *) {}
void UnityPluginUnload(void) {}

rm_ctx *rm_create(int hip_device) {
    rm_ctx *c = nullptr;
    try {
        if (hip_device < 0) { g_create_error = "negative device index"; return nullptr; }
        c = new (std::nothrow) rm_ctx();
        if (!c) { g_create_error = "out of memory"; return nullptr; }
        c->device = hip_device;
        load_defaults(c);
        return c;
    } catch (...) {
        delete c;
        try { g_create_error = "out of memory"; } catch (...) {}
        return nullptr;
    }
}

void rm_destroy(rm_ctx *c) {
    if (!c) return;
    if (c->dev_ready) {
        DeviceScope ds;
        (void)ds.enter(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        (void)wait_for_last(c);
        for (int k = 0; k < 2; k++) if (c->async_used[k]) (void)hipEventSynchronize(c->ev_done[k]);
        for (int i = 0; i < 6; i++) if (c->d_stage[i]) (void)hipFree(c->d_stage[i]);
        for (int i = 0; i < 6; i++) if (c->d_stage_b[i]) (void)hipFree(c->d_stage_b[i]);
        for (int k = 0; k < 2; k++) if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
        if (c->stream_b) { (void)hipStreamSynchronize(c->stream_b); (void)hipStreamDestroy(c->stream_b); }
        if (c->d_prims) (void)hipFree(c->d_prims);
        if (c->d_views) (void)hipFree(c->d_views);
        if (c->d_leaf_stats) (void)hipFree(c->d_leaf_stats);
        if (c->d_stats) (void)hipFree(c->d_stats);
        if (c->d_partials) (void)hipFree(c->d_partials);
        if (c->q.buf) (void)hipFree(c->q.buf);
        if (c->ev_last) (void)hipEventDestroy(c->ev_last);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

const char *rm_last_error(rm_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int rm_set_float(rm_ctx *c, const char *name, float v) {
    RM_TRY
    if (!c || !name) return fail(c, RM_E_ARG, "rm_set_float: null argument");
    c->floats[name] = v;
    return RM_OK;
    RM_CATCH(c)
}
int rm_set_int(rm_ctx *c, const char *name, int v) {
    RM_TRY
    // Unity 2018 keeps SetInt values in the material's float table
    if (!c || !name) return fail(c, RM_E_ARG, "rm_set_int: null argument");
    c->floats[name] = (float)v;
    return RM_OK;
    RM_CATCH(c)
}
int rm_set_vector(rm_ctx *c, const char *name, const float v[4]) {
    RM_TRY
    if (!c || !name || !v) return fail(c, RM_E_ARG, "rm_set_vector: null argument");
    c->vectors[name] = {v[0], v[1], v[2], v[3]};
    return RM_OK;
    RM_CATCH(c)
}
int rm_set_matrix(rm_ctx *c, const char *name, const float m[16]) {
    RM_TRY
    if (!c || !name || !m) return fail(c, RM_E_ARG, "rm_set_matrix: null argument");
    mat4 a;
    std::memcpy(a.data(), m, sizeof(float) * 16);
    c->matrices[name] = a;
    return RM_OK;
    RM_CATCH(c)
}
int rm_get_float(rm_ctx *c, const char *name, float *v) {
    RM_TRY
    if (!c || !name || !v) return fail(c, RM_E_ARG, "rm_get_float: null argument");
    auto it = c->floats.find(name);
    if (it == c->floats.end()) return fail(c, RM_E_ARG, std::string("no float property ") + name);
    *v = it->second;
    return RM_OK;
    RM_CATCH(c)
}
int rm_get_int(rm_ctx *c, const char *name, int *v) {
    RM_TRY
    float f;
    if (!v) return fail(c, RM_E_ARG, "rm_get_int: null argument");
    int rc = rm_get_float(c, name, &f);
    if (rc == RM_OK) *v = (int)f;
    return rc;
    RM_CATCH(c)
}
int rm_get_vector(rm_ctx *c, const char *name, float v[4]) {
    RM_TRY
    if (!c || !name || !v) return fail(c, RM_E_ARG, "rm_get_vector: null argument");
    auto it = c->vectors.find(name);
    if (it == c->vectors.end()) return fail(c, RM_E_ARG, std::string("no vector property ") + name);
    std::memcpy(v, it->second.data(), sizeof(float) * 4);
    return RM_OK;
    RM_CATCH(c)
}
int rm_get_matrix(rm_ctx *c, const char *name, float m[16]) {
    RM_TRY
    if (!c || !name || !m) return fail(c, RM_E_ARG, "rm_get_matrix: null argument");
    auto it = c->matrices.find(name);
    if (it == c->matrices.end()) return fail(c, RM_E_ARG, std::string("no matrix property ") + name);
    std::memcpy(m, it->second.data(), sizeof(float) * 16);
    return RM_OK;
    RM_CATCH(c)
}
int rm_has_property(rm_ctx *c, const char *name) {
    RM_TRY
    if (!c || !name) return 0;
    for (const char *d : kDeclared) if (std::strcmp(d, name) == 0) return 1;
    return 0;
    RM_CATCH(c)
}

int rm_set_texture_device(rm_ctx *c, const char *name, const void *d_texels, int w, int h) {
    RM_TRY
    if (!c || !name) return fail(c, RM_E_ARG, "rm_set_texture_device: null argument");
    if (std::strcmp(name, "_ShadowMap") != 0) return RM_OK;       // other texture slots are stored nowhere: the blit takes _MainTex and the depth texture as arguments
    if (d_texels && (w <= 0 || h <= 0 || w > 32768 || h > 32768)) return fail(c, RM_E_ARG, "rm_set_texture_device: bad size");
    if ((uintptr_t)d_texels & 3u) return fail(c, RM_E_ARG, "rm_set_texture_device: the depth plane must be 4-byte aligned");
    c->shadow_map = (const float *)d_texels;
    c->shadow_w = d_texels ? w : 0; c->shadow_h = d_texels ? h : 0;
    return RM_OK;
    RM_CATCH(c)
}

int rm_set_primitives(rm_ctx *c, const rm_primitive *prims, int n) {
    RM_TRY
    if (!c || n < 0 || (n > 0 && !prims)) return fail(c, RM_E_ARG, "rm_set_primitives: bad argument");
    if (n > RM_MAX_PRIMS) return fail(c, RM_E_RANGE, "more than RM_MAX_PRIMS primitives");
    for (int i = 0; i < n; i++)
        if (prims[i].type < RM_SPHERE || prims[i].type > RM_TORUS || (prims[i].op & ~(RM_OP_SMAX | RM_OP_NEGATE)) != 0)
            return fail(c, RM_E_ARG, "rm_set_primitives: unknown primitive type or operator");
    c->prims.assign(prims, prims + n);
    c->prims_dirty = true;
    return RM_OK;
    RM_CATCH(c)
}

int rm_set_kernel(rm_ctx *c, int variant) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (variant < RM_KERNEL_AUTO || variant > RM_KERNEL_COOP) return fail(c, RM_E_ARG, "unknown kernel variant");
    c->kernel = variant;
    return RM_OK;
    RM_CATCH(c)
}

int rm_band_rows(int h, int band_rows, int part, int n_parts) {
    if (h <= 0 || band_rows <= 0 || n_parts <= 0 || part < 0 || part >= n_parts) return RM_E_ARG;
    int rows = 0;
    int n_bands = (h + band_rows - 1) / band_rows;
    for (int b = part; b < n_bands; b += n_parts) {
        int y0 = b * band_rows, y1 = y0 + band_rows;
        if (y1 > h) y1 = h;
        rows += y1 - y0;
    }
    return rows;
}

static int blit_device_common(rm_ctx *c, const void *d_src, const void *d_sdepth, void *d_rgba, void *d_depth, void *d_hit,
                              void *d_steps, int w, int h, int band_rows, int part, int n_parts, void *hip_stream) {
    int rc = check_size(c, w, h);
    if (rc) return rc;
    int rows = rm_band_rows(h, band_rows, part, n_parts);
    if (rows < 0) return fail(c, RM_E_ARG, "rm_blit_bands_device: bad band arguments");
    if (rows == 0) return RM_OK;
    if (d_steps && ((uintptr_t)d_steps & 3u)) return fail(c, RM_E_ARG, "step-count plane must be 4-byte aligned");
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    if ((rc = ensure_device(c))) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    RmParams U;
    if ((rc = build_params(c, &U, stream))) return rc;
    U.w = w; U.h = h; U.rows = rows;
    U.band_rows = band_rows; U.part = part; U.n_parts = n_parts;
    U.mode = RM_MODE_FRAG;
    U.src = (const uint8_t *)d_src; U.sdepth = (const float *)d_sdepth;
    U.dst_rgba = (uint8_t *)d_rgba; U.dst_depth = (float *)d_depth; U.dst_hit = (uint8_t *)d_hit;
    U.dst_steps = (uint32_t *)d_steps;
    return launch(c, U, stream, own_mem(c), true);
}

int rm_blit_bands_device(rm_ctx *c, const void *d_src, const void *d_sdepth, void *d_rgba, void *d_depth, void *d_hit,
                         int w, int h, int band_rows, int part, int n_parts, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    return blit_device_common(c, d_src, d_sdepth, d_rgba, d_depth, d_hit, nullptr, w, h, band_rows, part, n_parts, hip_stream);
    RM_CATCH(c)
}

int rm_blit_device(rm_ctx *c, const void *d_src, const void *d_sdepth, void *d_rgba, void *d_depth, void *d_hit,
                   int w, int h, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    return blit_device_common(c, d_src, d_sdepth, d_rgba, d_depth, d_hit, nullptr, w, h, h > 0 ? h : 1, 0, 1, hip_stream);
    RM_CATCH(c)
}

int rm_blit_steps_device(rm_ctx *c, const void *d_src, const void *d_sdepth, void *d_rgba, void *d_depth, void *d_hit,
                         void *d_steps, int w, int h, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    return blit_device_common(c, d_src, d_sdepth, d_rgba, d_depth, d_hit, d_steps, w, h, h > 0 ? h : 1, 0, 1, hip_stream);
    RM_CATCH(c)
}

int rm_blit_device_noalloc(rm_ctx *c, const void *d_src, const void *d_sdepth, void *d_rgba, void *d_depth, void *d_hit,
                           int w, int h, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    int rc = check_size(c, w, h);
    if (rc) return rc;
    static_assert(queue_words(RM_SCRATCH_BM_WORDS) * 4 == RM_SCRATCH_QUEUE_BYTES, "scratch layout follows the queue layout");
    const size_t slot_cap = RM_SCRATCH_SLOT_BYTES / (4 * (size_t)rm_slot_words());
    if ((size_t)rm_tile_blocks(w, h, 1) > slot_cap)
        return fail(c, RM_E_RANGE, "rm_blit_device_noalloc: frame beyond the static scratch (at most 32768 blocks of 16x16 pixels, "
                                   "e.g. 3840x2160); use rm_blit_device");
    if (!c->prims.empty())
        return fail(c, RM_E_ARG, "rm_blit_device_noalloc: a primitive table lives in memory the library allocates; use rm_blit_device");
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    if ((rc = ensure_props(c))) return rc;
    DeviceScratch *D = nullptr;
    if ((rc = device_scratch(c, &D))) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    RmParams U;
    if ((rc = build_params(c, &U, stream))) return rc;
    U.w = w; U.h = h; U.rows = h;
    U.band_rows = h; U.part = 0; U.n_parts = 1;
    U.mode = RM_MODE_FRAG;
    U.src = (const uint8_t *)d_src; U.sdepth = (const float *)d_sdepth;
    U.dst_rgba = (uint8_t *)d_rgba; U.dst_depth = (float *)d_depth; U.dst_hit = (uint8_t *)d_hit;
    WorkMem M;
    M.q = &D->q; M.partials = &D->partials; M.partials_cap = &D->partials_cap; M.stats = D->stats; M.fixed = true;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        M.capturing = cs == hipStreamCaptureStatusActive;
    }
    rc = launch(c, U, stream, M, false);
    if (rc == RM_OK) {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        c->last_scratch_gen = ++D->generation;
    }
    return rc;
    RM_CATCH(c)
}

int rm_blit_poses_device(rm_ctx *c, int n_poses, const float *fcs, const float *c2ws, const float *cam_pos,
                         void *d_rgba, void *d_depth, void *d_hit, int w, int h, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (n_poses <= 0 || !fcs || !c2ws || !cam_pos) return fail(c, RM_E_ARG, "rm_blit_poses_device: bad argument");
    if (n_poses > 65535) return fail(c, RM_E_RANGE, "more than 65535 poses in one call");
    int rc = check_size(c, w, h);
    if (rc) return rc;
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    if ((rc = ensure_device(c))) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    RmParams U;
    if ((rc = build_params(c, &U, stream))) return rc;
    std::vector<RmView> views((size_t)n_poses);
    for (int i = 0; i < n_poses; i++) {
        mat4 fc, c2w;
        std::memcpy(fc.data(), fcs + (size_t)i * 16, 64);
        std::memcpy(c2w.data(), c2ws + (size_t)i * 16, 64);
        make_view(fc, c2w, cam_pos + (size_t)i * 3, &views[(size_t)i]);
    }
    // earlier launches (on whatever stream) may still read the previous views
    if ((rc = wait_for_last(c))) return rc;
    if ((rc = ensure_cap(c, &c->d_views, &c->d_views_cap, views.size() * sizeof(RmView)))) return rc;
    RM_HIP(c, hipMemcpy(c->d_views, views.data(), views.size() * sizeof(RmView), hipMemcpyHostToDevice));
    U.views = c->d_views;
    U.n_poses = n_poses;
    U.w = w; U.h = h; U.rows = h; U.band_rows = h;
    U.mode = RM_MODE_FRAG;
    U.dst_rgba = (uint8_t *)d_rgba; U.dst_depth = (float *)d_depth; U.dst_hit = (uint8_t *)d_hit;
    return launch(c, U, stream, own_mem(c), true);
    RM_CATCH(c)
}

static int blit_host_common(rm_ctx *c, const uint8_t *src, const float *sdepth, uint8_t *rgba, float *depth, uint8_t *hit,
                            uint16_t *steps, int w, int h) {
    int rc = check_size(c, w, h);
    if (rc) return rc;
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    if ((rc = ensure_device(c))) return rc;
    size_t n = (size_t)w * (size_t)h;
    // asynchronous host blits still in flight use the same staging set: let them deliver first
    for (int k = 0; k < 2; k++) if (c->async_used[k]) RM_HIP(c, hipEventSynchronize(c->ev_done[k]));
    // the staging buffers may still be read or written by this context's previous launch on another stream
    if ((rc = order_after_last(c, c->stream))) return rc;
    if (src)    { if ((rc = stage(c, 0, n * 4))) return rc; RM_HIP(c, hipMemcpyAsync(c->d_stage[0], src, n * 4, hipMemcpyHostToDevice, c->stream)); }
    if (sdepth) { if ((rc = stage(c, 1, n * 4))) return rc; RM_HIP(c, hipMemcpyAsync(c->d_stage[1], sdepth, n * 4, hipMemcpyHostToDevice, c->stream)); }
    if (rgba)  if ((rc = stage(c, 2, n * 4))) return rc;
    if (depth) if ((rc = stage(c, 3, n * 4))) return rc;
    if (hit)   if ((rc = stage(c, 4, n))) return rc;
    if (steps) if ((rc = stage(c, 5, n * 4))) return rc;
    rc = blit_device_common(c, src ? c->d_stage[0] : nullptr, sdepth ? c->d_stage[1] : nullptr,
                            rgba ? c->d_stage[2] : nullptr, depth ? c->d_stage[3] : nullptr, hit ? c->d_stage[4] : nullptr,
                            steps ? c->d_stage[5] : nullptr, w, h, h, 0, 1, c->stream);
    if (rc) return rc;
    if (rgba)  RM_HIP(c, hipMemcpyAsync(rgba, c->d_stage[2], n * 4, hipMemcpyDeviceToHost, c->stream));
    if (depth) RM_HIP(c, hipMemcpyAsync(depth, c->d_stage[3], n * 4, hipMemcpyDeviceToHost, c->stream));
    if (hit)   RM_HIP(c, hipMemcpyAsync(hit, c->d_stage[4], n, hipMemcpyDeviceToHost, c->stream));
    if (steps) RM_HIP(c, hipMemcpyAsync(steps, c->d_stage[5], n * 4, hipMemcpyDeviceToHost, c->stream));
    RM_HIP(c, hipStreamSynchronize(c->stream));
    return RM_OK;
}

// first use of rm_blit_async: the second set's stream and the two events the host waits on
static int ensure_async(rm_ctx *c) {
    if (!c->stream_b) RM_HIP(c, hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
    // the host waits on these and then reads what the copies wrote: they keep the system-scope fence
    for (int k = 0; k < 2; k++) if (!c->ev_done[k]) RM_HIP(c, hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming));
    return RM_OK;
}

int rm_blit_async(rm_ctx *c, const uint8_t *src, const float *sdepth, uint8_t *rgba, float *depth, uint8_t *hit, int w, int h,
                  long long *ticket) {
    RM_TRY
    if (!c || !ticket) return fail(c, RM_E_ARG, "rm_blit_async: null argument");
    int rc = check_size(c, w, h);
    if (rc) return rc;
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    if ((rc = ensure_device(c))) return rc;
    if ((rc = ensure_async(c))) return rc;
    // Calls alternate between the two sets.  Everything of one call -- uploads, march, copies back -- is queued on its
    // set's stream in that order, so a set's buffers are reused in stream order without an event; across the sets only
    // the marches are ordered (every launch waits for the context's previous one, launch()).  The copies back of call N
    // therefore run beside the uploads and the march of call N + 1.
    const int k = (int)(c->async_issued & 1);
    void **st = k ? c->d_stage_b : c->d_stage;
    size_t *cap = k ? c->d_stage_b_cap : c->d_stage_cap;
    hipStream_t s = k ? c->stream_b : c->stream;
    const size_t n = (size_t)w * (size_t)h;
    const size_t need[5] = {src ? n * 4 : 0, sdepth ? n * 4 : 0, rgba ? n * 4 : 0, depth ? n * 4 : 0, hit ? n : 0};
    bool grow = false;
    for (int i = 0; i < 5; i++) grow = grow || need[i] > cap[i];
    if (grow) {
        // growing frees the old buffers: whatever still uses this set has to be through
        RM_HIP(c, hipStreamSynchronize(s));
        if ((rc = wait_for_last(c))) return rc;
        for (int i = 0; i < 5; i++) if (need[i] && (rc = ensure_cap(c, &st[i], &cap[i], need[i]))) return rc;
    }
    if (src) RM_HIP(c, hipMemcpyAsync(st[0], src, n * 4, hipMemcpyHostToDevice, s));
    if (sdepth) RM_HIP(c, hipMemcpyAsync(st[1], sdepth, n * 4, hipMemcpyHostToDevice, s));
    rc = blit_device_common(c, src ? st[0] : nullptr, sdepth ? st[1] : nullptr, rgba ? st[2] : nullptr, depth ? st[3] : nullptr,
                            hit ? st[4] : nullptr, nullptr, w, h, h, 0, 1, s);
    if (rc) return rc;
    if (rgba)  RM_HIP(c, hipMemcpyAsync(rgba, st[2], n * 4, hipMemcpyDeviceToHost, s));
    if (depth) RM_HIP(c, hipMemcpyAsync(depth, st[3], n * 4, hipMemcpyDeviceToHost, s));
    if (hit)   RM_HIP(c, hipMemcpyAsync(hit, st[4], n, hipMemcpyDeviceToHost, s));
    RM_HIP(c, hipEventRecord(c->ev_done[k], s));
    c->async_used[k] = true;
    *ticket = c->async_issued++;
    return RM_OK;
    RM_CATCH(c)
}

int rm_blit_wait(rm_ctx *c, long long ticket) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (ticket < 0 || ticket >= c->async_issued) return fail(c, RM_E_ARG, "rm_blit_wait: no such ticket");
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    // (if the set has been reused by ticket + 2, ticket + 4, ... this waits for the latest of those calls: the set's stream runs
    // them in order, so that is later, never earlier, than asked)
    RM_HIP(c, hipEventSynchronize(c->ev_done[ticket & 1]));
    return RM_OK;
    RM_CATCH(c)
}

void *rm_host_alloc(rm_ctx *c, size_t bytes) {
    try {
        if (!c || bytes == 0) { (void)fail(c, RM_E_ARG, "rm_host_alloc: bad argument"); return nullptr; }
        DeviceScope ds;
        if (ds.enter(c->device) != hipSuccess) { (void)fail(c, RM_E_DEVICE, "rm_host_alloc: cannot select the context's device"); return nullptr; }
        void *p = nullptr;
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) { (void)fail(c, RM_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); return nullptr; }
        return p;
    } catch (...) {
        return nullptr;
    }
}

void rm_host_free(rm_ctx *c, void *p) {
    if (!c || !p) return;
    DeviceScope ds;
    if (ds.enter(c->device) == hipSuccess) (void)hipHostFree(p);
}

int rm_blit(rm_ctx *c, const uint8_t *src, const float *sdepth, uint8_t *rgba, float *depth, uint8_t *hit, int w, int h) {
    RM_TRY
    if (!c) return RM_E_ARG;
    return blit_host_common(c, src, sdepth, rgba, depth, hit, nullptr, w, h);
    RM_CATCH(c)
}

int rm_blit_steps(rm_ctx *c, const uint8_t *src, const float *sdepth, uint8_t *rgba, float *depth, uint8_t *hit,
                  uint16_t *steps, int w, int h) {
    RM_TRY
    if (!c) return RM_E_ARG;
    return blit_host_common(c, src, sdepth, rgba, depth, hit, steps, w, h);
    RM_CATCH(c)
}

static int depth_pass_device(rm_ctx *c, int mode, const float corners[12], const float forward[3],
                             void *d_depth, void *d_hit, int w, int h, void *hip_stream,
                             int band_rows = 0, int part = 0, int n_parts = 1) {
    RM_TRY
    int rc = check_size(c, w, h);
    if (rc) return rc;
    if (band_rows <= 0) { band_rows = h; part = 0; n_parts = 1; }      // the whole map
    const int rows = rm_band_rows(h, band_rows, part, n_parts);
    if (rows < 0) return fail(c, RM_E_ARG, "depth pass: bad band arguments");
    if (rows == 0) return RM_OK;
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    if ((rc = ensure_device(c))) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    RmParams U;
    if ((rc = build_params(c, &U, stream))) return rc;
    U.w = w; U.h = h; U.rows = rows;
    U.band_rows = band_rows; U.part = part; U.n_parts = n_parts;
    U.mode = mode;
    if (mode == RM_MODE_DEPTH_ORTHO) {
        std::memcpy(U.ortho_c, corners, sizeof(float) * 12);
        std::memcpy(U.ortho_fwd, forward, sizeof(float) * 3);
    }
    U.dst_depth = (float *)d_depth; U.dst_hit = (uint8_t *)d_hit;
    return launch(c, U, stream, own_mem(c), true);
    RM_CATCH(c)
}

static int depth_pass_host(rm_ctx *c, int mode, const float corners[12], const float forward[3], float *depth, int w, int h) {
    RM_TRY
    if (!depth) return fail(c, RM_E_ARG, "null destination");
    int rc = check_size(c, w, h);
    if (rc) return rc;
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    if ((rc = ensure_device(c))) return rc;
    size_t n = (size_t)w * (size_t)h;
    for (int k = 0; k < 2; k++) if (c->async_used[k]) RM_HIP(c, hipEventSynchronize(c->ev_done[k]));     // rm_blit_async calls in flight share the staging set
    if ((rc = stage(c, 3, n * 4))) return rc;
    if ((rc = depth_pass_device(c, mode, corners, forward, c->d_stage[3], nullptr, w, h, c->stream))) return rc;
    RM_HIP(c, hipMemcpyAsync(depth, c->d_stage[3], n * 4, hipMemcpyDeviceToHost, c->stream));
    RM_HIP(c, hipStreamSynchronize(c->stream));
    return RM_OK;
    RM_CATCH(c)
}

int rm_shadowmap_render_device(rm_ctx *c, void *d_depth, void *d_hit, int w, int h, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    return depth_pass_device(c, RM_MODE_DEPTH_PERSP, nullptr, nullptr, d_depth, d_hit, w, h, hip_stream);
    RM_CATCH(c)
}
int rm_shadowmap_render(rm_ctx *c, float *depth, int w, int h) {
    RM_TRY
    if (!c) return RM_E_ARG;
    return depth_pass_host(c, RM_MODE_DEPTH_PERSP, nullptr, nullptr, depth, w, h);
    RM_CATCH(c)
}
int rm_shadowmap_directional_device(rm_ctx *c, const float corners[12], const float forward[3], void *d_depth,
                                    void *d_hit, int w, int h, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (!corners || !forward) return fail(c, RM_E_ARG, "null corners/forward");
    return depth_pass_device(c, RM_MODE_DEPTH_ORTHO, corners, forward, d_depth, d_hit, w, h, hip_stream);
    RM_CATCH(c)
}
int rm_shadowmap_directional_bands_device(rm_ctx *c, const float corners[12], const float forward[3], void *d_depth,
                                          void *d_hit, int w, int h, int band_rows, int part, int n_parts, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (!corners || !forward) return fail(c, RM_E_ARG, "null corners/forward");
    if (band_rows <= 0) return fail(c, RM_E_ARG, "rm_shadowmap_directional_bands_device: bad band arguments");
    return depth_pass_device(c, RM_MODE_DEPTH_ORTHO, corners, forward, d_depth, d_hit, w, h, hip_stream, band_rows, part, n_parts);
    RM_CATCH(c)
}
int rm_shadowmap_directional(rm_ctx *c, const float corners[12], const float forward[3], float *depth, int w, int h) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (!corners || !forward) return fail(c, RM_E_ARG, "null corners/forward");
    return depth_pass_host(c, RM_MODE_DEPTH_ORTHO, corners, forward, depth, w, h);
    RM_CATCH(c)
}

static int pack_common(rm_ctx *c, const void *in, void *out, size_t n_pixels, bool unpack, void *hip_stream) {
    if (!c) return RM_E_ARG;
    if (!in || !out) return fail(c, RM_E_ARG, "rm_pack_rgb_device: null buffer");
    if (n_pixels % 4 != 0) return fail(c, RM_E_ARG, "rm_pack_rgb_device: pixel count must be a multiple of 4");
    if (((uintptr_t)in | (uintptr_t)out) & 3u) return fail(c, RM_E_ARG, "rm_pack_rgb_device: buffers must be 4-byte aligned");
    if (!unpack && ((uintptr_t)in & 15u)) return fail(c, RM_E_ARG, "rm_pack_rgb_device: RGBA8 buffer must be 16-byte aligned");
    if (unpack && ((uintptr_t)out & 15u)) return fail(c, RM_E_ARG, "rm_unpack_rgb_device: RGBA8 buffer must be 16-byte aligned");
    int rc;
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    if ((rc = ensure_device(c))) return rc;
    RM_HIP(c, rm_pack_rgb(in, out, n_pixels, unpack, (hipStream_t)hip_stream));
    return RM_OK;
}
int rm_pack_rgb_device(rm_ctx *c, const void *d_rgba8, void *d_rgb8, size_t n_pixels, void *hip_stream) {
    RM_TRY
    return pack_common(c, d_rgba8, d_rgb8, n_pixels, false, hip_stream);
    RM_CATCH(c)
}
int rm_unpack_rgb_device(rm_ctx *c, const void *d_rgb8, void *d_rgba8, size_t n_pixels, void *hip_stream) {
    RM_TRY
    return pack_common(c, d_rgb8, d_rgba8, n_pixels, true, hip_stream);
    RM_CATCH(c)
}

size_t rm_tile_record_bytes(void) { return (size_t)rm_record_words() * 4; }

int rm_compact_frames_device(rm_ctx *c, const void *d_rgba8, const void *d_hit, int w, int h, int n_frames,
                             void *d_records, void *d_count, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (!d_rgba8 || !d_hit || !d_records || !d_count || n_frames <= 0) return fail(c, RM_E_ARG, "rm_compact_frames_device: bad argument");
    if (((uintptr_t)d_rgba8 | (uintptr_t)d_records | (uintptr_t)d_count) & 3u) return fail(c, RM_E_ARG, "rm_compact_frames_device: buffers must be 4-byte aligned");
    int rc = check_size(c, w, h);
    if (rc) return rc;
    const unsigned long long tiles = (unsigned long long)((w + 7) / 8) * (unsigned long long)((h + 7) / 8) * (unsigned long long)n_frames;
    if (tiles > 0x7fffffffull) return fail(c, RM_E_RANGE, "rm_compact_frames_device: more than 2^31 tiles in one call");
    // The records hold the tiles with a hit pixel; everything else is rebuilt from the background.  The step-count term,
    // getscatter and the light integration along the ray (shader:203,206-212,129-140) also colour miss pixels, so with any
    // of them set a frame is not "background outside the hit tiles" and the transport would silently lose those pixels.
    if (getf(c, "_RmStepHeat", 0.0f) != 0.0f || getf(c, "_RmScatter", 0.0f) != 0.0f || getf(c, "_RmVolumetricLight", 0.0f) != 0.0f)
        return fail(c, RM_E_ARG, "rm_compact_frames_device: _RmStepHeat, _RmScatter and _RmVolumetricLight colour miss pixels too; "
                                 "such frames are not the background outside their hit tiles (send them whole: rm_pack_rgb_device)");
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    RM_HIP(c, rm_compact_tiles(d_rgba8, d_hit, w, h, (unsigned)n_frames, d_records, d_count, (hipStream_t)hip_stream));
    return RM_OK;
    RM_CATCH(c)
}

int rm_expand_frames_device(rm_ctx *c, const void *d_records, uint32_t n_records, const void *d_src_rgba8, void *d_dst_rgba8,
                            int w, int h, int n_frames, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (!d_dst_rgba8 || n_frames <= 0 || (n_records && !d_records)) return fail(c, RM_E_ARG, "rm_expand_frames_device: bad argument");
    if (((uintptr_t)d_dst_rgba8 | (uintptr_t)d_records) & 3u) return fail(c, RM_E_ARG, "rm_expand_frames_device: buffers must be 4-byte aligned");
    int rc = check_size(c, w, h);
    if (rc) return rc;
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    const size_t px = (size_t)w * (size_t)h;
    if (d_src_rgba8) {
        // miss pixels show _MainTex with the alpha the blit writes (fixed4(col, 1), shader:214)
        if ((uintptr_t)d_src_rgba8 & 3u) return fail(c, RM_E_ARG, "rm_expand_frames_device: buffers must be 4-byte aligned");
        RM_HIP(c, rm_fill_from_src(d_src_rgba8, d_dst_rgba8, px, (unsigned)n_frames, stream));
    } else {
        vec4 bg = getv(c, "_BackgroundColor");
        const uint32_t bg8 = unorm8(bg[0]) | (unorm8(bg[1]) << 8) | (unorm8(bg[2]) << 16) | 0xFF000000u;
        RM_HIP(c, hipMemsetD32Async((hipDeviceptr_t)d_dst_rgba8, (int)bg8, px * (size_t)n_frames, stream));
    }
    RM_HIP(c, rm_expand_tiles(d_records, n_records, w, h, (unsigned)n_frames, d_dst_rgba8, stream));
    return RM_OK;
    RM_CATCH(c)
}

int rm_update_frames_device(rm_ctx *c, const void *d_old_records, uint32_t n_old, const void *d_new_records, uint32_t n_new,
                            const void *d_src_rgba8, void *d_dst_rgba8, int w, int h, int n_frames, void *hip_stream) {
    RM_TRY
    if (!c) return RM_E_ARG;
    if (!d_dst_rgba8 || n_frames <= 0 || (n_old && !d_old_records) || (n_new && !d_new_records))
        return fail(c, RM_E_ARG, "rm_update_frames_device: bad argument");
    if (((uintptr_t)d_dst_rgba8 | (uintptr_t)d_old_records | (uintptr_t)d_new_records | (uintptr_t)d_src_rgba8) & 3u)
        return fail(c, RM_E_ARG, "rm_update_frames_device: buffers must be 4-byte aligned");
    int rc = check_size(c, w, h);
    if (rc) return rc;
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    vec4 bg = getv(c, "_BackgroundColor");
    const uint32_t bg8 = unorm8(bg[0]) | (unorm8(bg[1]) << 8) | (unorm8(bg[2]) << 16) | 0xFF000000u;
    RM_HIP(c, rm_restore_tiles(d_old_records, n_old, w, h, (unsigned)n_frames, d_src_rgba8, bg8, d_dst_rgba8, stream));
    RM_HIP(c, rm_expand_tiles(d_new_records, n_new, w, h, (unsigned)n_frames, d_dst_rgba8, stream));
    return RM_OK;
    RM_CATCH(c)
}

int rm_debug_counter_slots(rm_ctx *c, uint32_t *out, int max_words, int *words_per_slot, int *n_slots) {
    RM_TRY
    if (!c || !c->have_last) return fail(c, RM_E_ARG, "rm_debug_counter_slots: no blit has run");
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    int rc = scratch_still_ours(c);
    if (rc) return rc;
    if ((rc = wait_for_last(c))) return rc;
    size_t words = (size_t)c->last_blocks * rm_slot_words();
    if (words_per_slot) *words_per_slot = (int)rm_slot_words();
    if (n_slots) *n_slots = (int)c->last_blocks;
    if (out && max_words > 0) {
        if (words > (size_t)max_words) words = (size_t)max_words;
        RM_HIP(c, hipMemcpy(out, c->last_partials, words * 4, hipMemcpyDeviceToHost));
    }
    return RM_OK;
    RM_CATCH(c)
}

int rm_get_leaf_stats(rm_ctx *c, uint64_t *visited, uint64_t *possible) {
    RM_TRY
    if (!c || !visited || !possible) return fail(c, RM_E_ARG, "rm_get_leaf_stats: null argument");
    *visited = *possible = 0;
    if (!c->have_last || c->last_leaf_waves == 0)
        return fail(c, RM_E_ARG, "rm_get_leaf_stats: the last call did not run the table kernel with exact culling");
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    int rc = wait_for_last(c);
    if (rc) return rc;
    std::vector<uint32_t> h((size_t)c->last_leaf_waves * 2);
    RM_HIP(c, hipMemcpy(h.data(), c->d_leaf_stats, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t v = 0, p = 0;
    for (size_t i = 0; i < h.size(); i += 2) { v += h[i]; p += h[i + 1]; }
    *visited = v; *possible = p;
    return RM_OK;
    RM_CATCH(c)
}

int rm_get_stats(rm_ctx *c, rm_stats *out) {
    RM_TRY
    if (!c || !out) return fail(c, RM_E_ARG, "rm_get_stats: null argument");
    if (!c->have_last) return fail(c, RM_E_ARG, "rm_get_stats: no blit has run");
    DeviceScope ds;
    RM_HIP(c, ds.enter(c->device));
    {
        int rc = scratch_still_ours(c);
        if (rc) return rc;
    }
    // The reduction runs on the context's own stream, ordered behind the last launch by its event: the stream
    // that launch was given is the caller's and may be gone by now.  A launch of the entry point that
    // allocates nothing leaves no event; its stream is used, and must still exist.
    hipStream_t s = c->last_stream;
    if (c->last_has_event) {
        s = c->stream;
        RM_HIP(c, hipStreamWaitEvent(s, c->ev_last, 0));
    }
    RM_HIP(c, rm_reduce_stats(c->last_partials, c->last_blocks, c->last_stats, s));
    RM_HIP(c, hipStreamSynchronize(s));
    unsigned long long v[4];
    RM_HIP(c, hipMemcpy(v, c->last_stats, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    out->n_pixels = c->last_pixels;
    out->n_hit = v[0];
    out->sum_steps_p = v[1];
    out->sum_steps_s = v[2];
    return RM_OK;
    RM_CATCH(c)
}

}  // extern "C"
