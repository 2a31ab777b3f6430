// rmhip.hpp -- header-only C++ host mirror of the Unity objects the reference's scripts use to
// drive the march: Material (property bag), RenderTexture, Graphics::Blit, and the three
// script classes (Assets/Raymarching.cs, Assets/ShadowMap.cs, Assets/ShadowMapDirectional.cs)
// with their method names.  Everything forwards to the C ABI in rmhip.h; no arithmetic of the
// march lives here.  Errors become std::runtime_error carrying rm_last_error().
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rmhip.h"

namespace rmhip {

struct Vector3 { float x = 0, y = 0, z = 0; };
struct Vector4 { float x = 0, y = 0, z = 0, w = 0; };
struct Quaternion { float x = 0, y = 0, z = 0, w = 1; };

inline Vector3 rotate(const Quaternion &q, const Vector3 &v) {   // Quaternion * Vector3
    float x2 = q.x * 2, y2 = q.y * 2, z2 = q.z * 2;
    float xx = q.x * x2, yy = q.y * y2, zz = q.z * z2, xy = q.x * y2, xz = q.x * z2, yz = q.y * z2;
    float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    return {(1 - (yy + zz)) * v.x + (xy - wz) * v.y + (xz + wy) * v.z,
            (xy + wz) * v.x + (1 - (xx + zz)) * v.y + (yz - wx) * v.z,
            (xz - wy) * v.x + (yz + wx) * v.y + (1 - (xx + yy)) * v.z};
}

// sixteen floats in Unity's field order m00,m10,m20,m30,m01,... (column-major)
struct Matrix4x4 {
    std::array<float, 16> m{};
    static Matrix4x4 identity() { Matrix4x4 r; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1; return r; }
    void SetRow(int i, const Vector4 &v) { m[i] = v.x; m[i + 4] = v.y; m[i + 8] = v.z; m[i + 12] = v.w; }
    void SetRow(int i, const Vector3 &v) { SetRow(i, Vector4{v.x, v.y, v.z, 0}); }
    Vector4 GetRow(int i) const { return {m[i], m[i + 4], m[i + 8], m[i + 12]}; }
};

struct Transform {
    Vector3 position; Quaternion rotation; Vector3 lossyScale{1, 1, 1};
    Vector3 forward() const { return rotate(rotation, Vector3{0, 0, 1}); }
};

struct Camera {
    Transform transform;
    float fieldOfView = 60, aspect = 16.0f / 9.0f, nearClipPlane = 0.3f, farClipPlane = 1000;
    // TRS(position, rotation, 1) with the Z column negated (camera space looks down -Z)
    Matrix4x4 cameraToWorldMatrix() const {
        Vector3 c0 = rotate(transform.rotation, {1, 0, 0}), c1 = rotate(transform.rotation, {0, 1, 0}),
                c2 = rotate(transform.rotation, {0, 0, 1});
        Matrix4x4 r = Matrix4x4::identity();
        r.SetRow(0, Vector4{c0.x, c1.x, -c2.x, transform.position.x});
        r.SetRow(1, Vector4{c0.y, c1.y, -c2.y, transform.position.y});
        r.SetRow(2, Vector4{c0.z, c1.z, -c2.z, transform.position.z});
        return r;
    }
};

class Material {
public:
    explicit Material(int hip_device = 0) : ctx_(rm_create(hip_device)) {
        if (!ctx_) throw std::runtime_error(std::string("rm_create: ") + rm_last_error(nullptr));
    }
    ~Material() { rm_destroy(ctx_); }
    Material(const Material &) = delete;
    Material &operator=(const Material &) = delete;

    void SetFloat(const char *n, float v) { ck(rm_set_float(ctx_, n, v)); }
    void SetInt(const char *n, int v) { ck(rm_set_int(ctx_, n, v)); }
    void SetVector(const char *n, const Vector4 &v) { float a[4] = {v.x, v.y, v.z, v.w}; ck(rm_set_vector(ctx_, n, a)); }
    void SetMatrix(const char *n, const Matrix4x4 &m) { ck(rm_set_matrix(ctx_, n, m.m.data())); }
    float GetFloat(const char *n) { float v; ck(rm_get_float(ctx_, n, &v)); return v; }
    int GetInt(const char *n) { int v; ck(rm_get_int(ctx_, n, &v)); return v; }
    Vector4 GetVector(const char *n) { float a[4]; ck(rm_get_vector(ctx_, n, a)); return {a[0], a[1], a[2], a[3]}; }
    Matrix4x4 GetMatrix(const char *n) { Matrix4x4 m; ck(rm_get_matrix(ctx_, n, m.m.data())); return m; }
    bool HasProperty(const char *n) { return rm_has_property(ctx_, n) != 0; }
    // Material.SetTexture for a texture that lives in device memory ("_ShadowMap": fp32 depths, row 0 = bottom row)
    void SetTextureDevice(const char *n, const void *d_texels, int w, int h) { ck(rm_set_texture_device(ctx_, n, d_texels, w, h)); }
    void set_kernel(int variant) { ck(rm_set_kernel(ctx_, variant)); }
    rm_stats stats() { rm_stats s; ck(rm_get_stats(ctx_, &s)); return s; }
    rm_ctx *handle() const { return ctx_; }
    void ck(int rc) const { if (rc != RM_OK) throw std::runtime_error(std::string("librmhip ") + std::to_string(rc) + ": " + rm_last_error(ctx_)); }

private:
    rm_ctx *ctx_;
};

// host-memory render target: RGBA8 colour, fp32 depth, hit mask; row 0 = bottom row
struct RenderTexture {
    int width, height;
    std::vector<uint8_t> color, hit;
    std::vector<float> depth;
    RenderTexture(int w, int h) : width(w), height(h), color((size_t)w * h * 4), hit((size_t)w * h), depth((size_t)w * h) {}
};

// page-locked host memory from the library: rm_blit copies into it at the full PCIe rate
template <class T> class PinnedBuffer {
public:
    PinnedBuffer(Material &m, size_t count) : mat_(m), n_(count), p_(static_cast<T *>(rm_host_alloc(m.handle(), count * sizeof(T)))) {
        if (!p_) throw std::runtime_error(std::string("rm_host_alloc: ") + rm_last_error(m.handle()));
    }
    ~PinnedBuffer() { rm_host_free(mat_.handle(), p_); }
    PinnedBuffer(const PinnedBuffer &) = delete;
    PinnedBuffer &operator=(const PinnedBuffer &) = delete;
    T *data() { return p_; }
    size_t size() const { return n_; }
private:
    Material &mat_; size_t n_; T *p_;
};

struct Graphics {
    // Graphics.Blit into device memory, asynchronous on hip_stream (a hipStream_t); any output but rgba8 may be null
    static void BlitDevice(Material &mat, const void *d_src_rgba8, const void *d_scene_depth, void *d_rgba8, void *d_depth, void *d_hit,
                           int w, int h, void *hip_stream) {
        mat.ck(rm_blit_device(mat.handle(), d_src_rgba8, d_scene_depth, d_rgba8, d_depth, d_hit, w, h, hip_stream));
    }
    // the same for callers that must not allocate (graph capture, serving loops): working memory is a static device array
    static void BlitDeviceNoAlloc(Material &mat, const void *d_src_rgba8, const void *d_scene_depth, void *d_rgba8, void *d_depth,
                                  void *d_hit, int w, int h, void *hip_stream) {
        mat.ck(rm_blit_device_noalloc(mat.handle(), d_src_rgba8, d_scene_depth, d_rgba8, d_depth, d_hit, w, h, hip_stream));
    }
    // Graphics.Blit(src, dest, mat); src may be null (constant _BackgroundColor)
    static void Blit(const RenderTexture *src, RenderTexture &dest, Material &mat, const float *scene_depth = nullptr) {
        if (src && (src->width != dest.width || src->height != dest.height)) throw std::invalid_argument("src and dest sizes differ");
        mat.ck(rm_blit(mat.handle(), src ? src->color.data() : nullptr, scene_depth, dest.color.data(), dest.depth.data(),
                       dest.hit.data(), dest.width, dest.height));
    }
};

inline Matrix4x4 GetFrustumCorners(const Camera &cam) {      // Assets/Raymarching.cs:32-56
    const float deg2rad = 0.0174532924f;
    float t = (float)std::tan((double)(cam.fieldOfView * 0.5f * deg2rad)), a = t * cam.aspect;
    Matrix4x4 m = Matrix4x4::identity();
    m.SetRow(0, Vector3{-a, t, -1});
    m.SetRow(1, Vector3{a, t, -1});
    m.SetRow(2, Vector3{a, -t, -1});
    m.SetRow(3, Vector3{-a, -t, -1});
    return m;
}

class Raymarching {                                            // Assets/Raymarching.cs
public:
    Camera cam; Material &mat; std::vector<Transform> transforms;
    Raymarching(const Camera &c, Material &m, std::vector<Transform> t) : cam(c), mat(m), transforms(std::move(t)) {}
    Matrix4x4 GetPositions() const { return rows([](const Transform &t) { return Vector4{t.position.x, t.position.y, t.position.z, 0}; }); }
    Matrix4x4 GetRotations() const { return rows([](const Transform &t) { return Vector4{t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w}; }); }
    Matrix4x4 GetScales() const { return rows([](const Transform &t) { return Vector4{t.lossyScale.x, t.lossyScale.y, t.lossyScale.z, 0}; }); }
    void OnRenderImage(const RenderTexture *src, RenderTexture &dest) {   // :85-95
        mat.SetMatrix("_FrustrumCorners", GetFrustumCorners(cam));
        mat.SetMatrix("_CameraInvViewMatrix", cam.cameraToWorldMatrix());
        mat.SetMatrix("_Positions", GetPositions());
        mat.SetMatrix("_Rotations", GetRotations());
        mat.SetMatrix("_Scales", GetScales());
        mat.SetVector("_WorldSpaceCameraPos", {cam.transform.position.x, cam.transform.position.y, cam.transform.position.z, 0});
        mat.SetVector("_ProjectionParams", {1, cam.nearClipPlane, cam.farClipPlane, 1 / cam.farClipPlane});
        Graphics::Blit(src, dest, mat);
    }
private:
    template <class F> Matrix4x4 rows(F pick) const {
        Matrix4x4 m = Matrix4x4::identity();
        for (size_t i = 0; i < transforms.size() && i < 4; i++) m.SetRow((int)i, pick(transforms[i]));
        return m;
    }
};

class ShadowMap {                                              // Assets/ShadowMap.cs
public:
    Camera cam; Material &shadowMapMaterial;
    ShadowMap(const Camera &c, Material &m) : cam(c), shadowMapMaterial(m) {}
    void Render(std::vector<float> &shadowMap, int w = 100, int h = 100) {   // :50-56
        shadowMap.resize((size_t)w * h);
        shadowMapMaterial.SetMatrix("_FrustrumCorners", GetFrustumCorners(cam));
        shadowMapMaterial.SetMatrix("_CameraInvViewMatrix", cam.cameraToWorldMatrix());
        shadowMapMaterial.SetVector("_WorldSpaceCameraPos", {cam.transform.position.x, cam.transform.position.y, cam.transform.position.z, 0});
        shadowMapMaterial.ck(rm_shadowmap_render(shadowMapMaterial.handle(), shadowMap.data(), w, h));
    }
};

class ShadowMapDirectional {                                   // Assets/ShadowMapDirectional.cs
public:
    Transform transform; Material &material; std::array<Vector3, 4> corners{};
    ShadowMapDirectional(const Transform &t, Material &m) : transform(t), material(m) {}
    void Update() {                                            // :25-34
        const Vector3 q[4] = {{-0.5f, 0.5f, 0}, {0.5f, 0.5f, 0}, {0.5f, -0.5f, 0}, {-0.5f, -0.5f, 0}};
        for (int i = 0; i < 4; i++) {
            Vector3 c{q[i].x * transform.lossyScale.x, q[i].y * transform.lossyScale.y, q[i].z * transform.lossyScale.z};
            c = rotate(transform.rotation, c);
            corners[i] = {c.x + transform.position.x, c.y + transform.position.y, c.z + transform.position.z};
        }
    }
    void Render(std::vector<float> &shadowMap, int w, int h) {
        Update();
        shadowMap.resize((size_t)w * h);
        float c[12], f[3];
        for (int i = 0; i < 4; i++) { c[3 * i] = corners[i].x; c[3 * i + 1] = corners[i].y; c[3 * i + 2] = corners[i].z; }
        Vector3 fw = transform.forward();
        f[0] = fw.x; f[1] = fw.y; f[2] = fw.z;
        material.ck(rm_shadowmap_directional(material.handle(), c, f, shadowMap.data(), w, h));
    }
};

}  // namespace rmhip
