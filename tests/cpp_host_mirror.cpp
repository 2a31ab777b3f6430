// Exercises the C++ host mirror (include/rmhip.hpp) against librmhip.so without a GPU:
// property bag round trips and the script classes' matrices.  With argument "gpu" it also
// renders a small frame through Raymarching::OnRenderImage and prints the hit count.
#include <cmath>
#include <cstdio>
#include <cstring>
#include "../include/rmhip.hpp"

using namespace rmhip;

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv) {
    Material mat;
    CHECK(mat.GetFloat("_MaxSteps") == 400.0f);
    CHECK(mat.HasProperty("_FrustrumCorners") && !mat.HasProperty("_FrustumCorners"));
    Camera cam;
    cam.transform.position = {2, 1, -11.01f};
    cam.fieldOfView = 80;
    Matrix4x4 fc = GetFrustumCorners(cam);
    CHECK(std::fabs(fc.GetRow(0).y - 0.8390996f) < 1e-6f && fc.GetRow(0).z == -1 && fc.GetRow(3).x < 0 && fc.GetRow(3).y < 0);
    Matrix4x4 c2w = cam.cameraToWorldMatrix();
    CHECK(c2w.GetRow(2).z == -1 && c2w.GetRow(2).w == -11.01f);
    std::vector<Transform> ts(4);
    ts[0].position = {-0.584f, 0.049f, 0.602f}; ts[1].position = {-0.74f, 0.929f, 0};
    ts[2].position = {-7.87f, 0, 1.522f};       ts[3].position = {-3.074f, -0.087f, 0.08f};
    Raymarching rm(cam, mat, ts);
    mat.SetMatrix("_Positions", rm.GetPositions());
    Matrix4x4 back = mat.GetMatrix("_Positions");
    CHECK(std::memcmp(back.m.data(), rm.GetPositions().m.data(), 64) == 0);
    CHECK(back.m[2] == -7.87f && back.m[6] == 0 && back.m[10] == 1.522f);      // row 2 in column-major fields
    Transform t;
    t.position = {1, 2, 3}; t.lossyScale = {4, 2, 1};
    t.rotation = {0, std::sqrt(0.5f), 0, std::sqrt(0.5f)};                     // 90 degrees about y
    ShadowMapDirectional smd(t, mat);
    smd.Update();
    CHECK(std::fabs(smd.corners[0].x - 1) < 1e-6f && std::fabs(smd.corners[0].y - 3) < 1e-6f && std::fabs(smd.corners[0].z - 5) < 1e-6f);
    bool threw = false;
    try { mat.GetFloat("_NeverSet"); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    if (argc > 1 && std::strcmp(argv[1], "gpu") == 0) {
        RenderTexture rt(320, 180);
        rm.cam.aspect = 320.0f / 180.0f;
        mat.SetVector("_WorldSpaceLightPos0", {0, 0.5f, 0.8660254f, 0});
        rm.OnRenderImage(nullptr, rt);
        std::printf("hits %llu\n", (unsigned long long)mat.stats().n_hit);
    }
    std::printf("ok\n");
    return 0;
}
