// ShadowMapNative.cs -- plugin-backed counterparts of the project's two shadow-map components.
//  * ShadowMapNative: perspective depth pass from this camera (what ShadowMap.Render() set up:
//    its own corner rays and camera-to-world matrix, then a blit into `shadowMap`).
//  * ShadowMapDirectionalNative: orthographic depth pass over the oriented rectangle that
//    ShadowMapDirectional computes from its transform every Update().
// Both write raymarch() length per texel into a float array / RFloat texture.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

[RequireComponent(typeof(Camera))]
public class ShadowMapNative : MonoBehaviour
{
    public RaymarchingNative scene;      // supplies the primitive transforms
    public int width = 100, height = 100;
    public int hipDevice = 0;
    [HideInInspector] public Texture2D shadowMap;
    [HideInInspector] public Camera cam;

    RmMaterial native;
    float[] depth;

    void OnEnable() { cam = GetComponent<Camera>(); native = new RmMaterial(hipDevice); }
    void OnDisable() { if (native != null) { native.Dispose(); native = null; } }

    public void Render()
    {
        if (shadowMap == null || shadowMap.width != width || shadowMap.height != height)
        {
            shadowMap = new Texture2D(width, height, TextureFormat.RFloat, false, true);
            depth = new float[width * height];
        }
        float t = Mathf.Tan(cam.fieldOfView * 0.5f * Mathf.Deg2Rad), a = t * cam.aspect;
        var corners = Matrix4x4.identity;
        corners.SetRow(0, new Vector4(-a, t, -1f, 0f));
        corners.SetRow(1, new Vector4(a, t, -1f, 0f));
        corners.SetRow(2, new Vector4(a, -t, -1f, 0f));
        corners.SetRow(3, new Vector4(-a, -t, -1f, 0f));
        if (scene) scene.BindScene(native);
        native.SetMatrix("_FrustrumCorners", corners);
        native.SetMatrix("_CameraInvViewMatrix", cam.cameraToWorldMatrix);
        native.SetVector("_WorldSpaceCameraPos", cam.transform.position);
        native.SetVector("_ProjectionParams", new Vector4(1f, cam.nearClipPlane, cam.farClipPlane, 1f / cam.farClipPlane));
        var pin = GCHandle.Alloc(depth, GCHandleType.Pinned);
        try { RmHip.Check(native.Ctx, RmHip.rm_shadowmap_render(native.Ctx, pin.AddrOfPinnedObject(), width, height)); }
        finally { pin.Free(); }
        shadowMap.SetPixelData(depth, 0);
        shadowMap.Apply(false);
    }
}

public class ShadowMapDirectionalNative : MonoBehaviour
{
    public RaymarchingNative scene;      // supplies the primitive transforms
    public int width = 1024, height = 1024;
    public int hipDevice = 0;
    [HideInInspector] public Texture2D shadowMap;

    static readonly Vector3[] UnitQuad = {
        new Vector3(-0.5f, 0.5f, 0f), new Vector3(0.5f, 0.5f, 0f), new Vector3(0.5f, -0.5f, 0f), new Vector3(-0.5f, -0.5f, 0f) };

    RmMaterial native;
    float[] depth;
    readonly float[] corners = new float[12];

    void OnEnable() { native = new RmMaterial(hipDevice); }
    void OnDisable() { if (native != null) { native.Dispose(); native = null; } }

    void Update()
    {
        // world-space corners of the rectangle: unit quad x lossyScale, rotated, translated
        for (int i = 0; i < 4; i++)
        {
            Vector3 c = transform.rotation * Vector3.Scale(UnitQuad[i], transform.lossyScale) + transform.position;
            corners[3 * i] = c.x; corners[3 * i + 1] = c.y; corners[3 * i + 2] = c.z;
        }
    }

    public void Render()
    {
        Update();
        if (shadowMap == null || shadowMap.width != width || shadowMap.height != height)
        {
            shadowMap = new Texture2D(width, height, TextureFormat.RFloat, false, true);
            depth = new float[width * height];
        }
        if (scene) scene.BindScene(native);
        Vector3 f = transform.forward;
        var pin = GCHandle.Alloc(depth, GCHandleType.Pinned);
        try
        {
            RmHip.Check(native.Ctx, RmHip.rm_shadowmap_directional(native.Ctx, corners, new[] { f.x, f.y, f.z },
                                                                  pin.AddrOfPinnedObject(), width, height));
        }
        finally { pin.Free(); }
        shadowMap.SetPixelData(depth, 0);
        shadowMap.Apply(false);
    }
}
