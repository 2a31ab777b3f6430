// RaymarchingNative.cs -- plugin-backed counterpart of the project's Raymarching image effect.
// Attach to the camera INSTEAD of Raymarching.cs.  Where that script binds five matrices on
// Raymarching.mat and calls Graphics.Blit (its OnRenderImage), this one binds the same names
// on an rm_ctx and calls rm_blit; the marched frame comes back as RGBA8 and is shown with a
// plain texture blit.  The float sliders still live on the Unity material asset and are
// mirrored into the plugin each frame, so the inspector keeps working.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

[ExecuteInEditMode]
[RequireComponent(typeof(Camera))]
public class RaymarchingNative : MonoBehaviour
{
    public Material mat;                 // Raymarching.mat: source of the serialized floats
    public Transform[] transforms;       // the four "Position N" objects
    public Light sun;                    // directional light whose direction becomes _WorldSpaceLightPos0
    public int hipDevice = 0;

    static readonly string[] FloatNames = {
        "_MaxSteps", "_MaxDist", "_StepFactor", "_ContactThreshold", "_NormalSampleScale",
        "_FractalScale", "_FractalRotationX", "_FractalRotationY", "_FractalRotationZ",
        "_Reflections", "_RefractiveIndex", "_Glossiness" };

    Camera cam;
    RmMaterial native;
    Texture2D frame;
    IntPtr pixels, depth;          // page-locked buffers from rm_host_alloc: the copies back run at PCIe rate
    int frameBytes;

    void OnEnable()
    {
        cam = GetComponent<Camera>();
        native = new RmMaterial(hipDevice);
    }

    void OnDisable()
    {
        Release();
        if (native != null) { native.Dispose(); native = null; }
    }

    void Release()
    {
        if (native != null)
        {
            if (pixels != IntPtr.Zero) RmHip.rm_host_free(native.Ctx, pixels);
            if (depth != IntPtr.Zero) RmHip.rm_host_free(native.Ctx, depth);
        }
        pixels = IntPtr.Zero; depth = IntPtr.Zero;
    }

    void Ensure(int w, int h)
    {
        if (frame != null && frame.width == w && frame.height == h) return;
        Release();
        frame = new Texture2D(w, h, TextureFormat.RGBA32, false, true);   // linear: the project is in Gamma space
        frameBytes = w * h * 4;
        pixels = RmHip.rm_host_alloc(native.Ctx, (UIntPtr)(ulong)frameBytes);
        depth = RmHip.rm_host_alloc(native.Ctx, (UIntPtr)(ulong)frameBytes);
        if (pixels == IntPtr.Zero || depth == IntPtr.Zero) throw new InvalidOperationException("librmhip: " + RmHip.LastError(native.Ctx));
    }

    // un-normalised view-space rays through the four image corners, one per matrix row
    static Matrix4x4 CornerRays(Camera c)
    {
        float t = Mathf.Tan(c.fieldOfView * 0.5f * Mathf.Deg2Rad);
        Vector3 right = Vector3.right * t * c.aspect, up = Vector3.up * t, back = -Vector3.forward;
        var m = Matrix4x4.identity;
        m.SetRow(0, back - right + up);
        m.SetRow(1, back + right + up);
        m.SetRow(2, back + right - up);
        m.SetRow(3, back - right - up);
        return m;
    }

    Matrix4x4 Rows(Func<Transform, Vector4> pick)
    {
        var m = Matrix4x4.identity;
        for (int i = 0; i < Mathf.Min(4, transforms.Length); i++)
            if (transforms[i]) m.SetRow(i, pick(transforms[i]));
        return m;
    }

    // the scene half of the bindings (floats + the three transform matrices), also used by the shadow passes
    public void BindScene(RmMaterial m)
    {
        m.CopyFloatsFrom(mat, FloatNames);
        m.SetMatrix("_Positions", Rows(t => t.position));
        m.SetMatrix("_Rotations", Rows(t => new Vector4(t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w)));
        m.SetMatrix("_Scales", Rows(t => t.lossyScale));
    }

    void OnRenderImage(RenderTexture src, RenderTexture dest)
    {
        int w = src.width, h = src.height;
        Ensure(w, h);

        BindScene(native);
        native.SetMatrix("_FrustrumCorners", CornerRays(cam));
        native.SetMatrix("_CameraInvViewMatrix", cam.cameraToWorldMatrix);
        // built-ins Unity would have bound for a forward pass
        native.SetVector("_WorldSpaceCameraPos", cam.transform.position);
        native.SetVector("_ProjectionParams", new Vector4(1f, cam.nearClipPlane, cam.farClipPlane, 1f / cam.farClipPlane));
        if (sun) native.SetVector("_WorldSpaceLightPos0", -sun.transform.forward);
        Color bg = cam.backgroundColor;
        native.SetVector("_BackgroundColor", new Vector4(bg.r, bg.g, bg.b, 1f));

        // no meshes in front of the march: no source image, no depth texture
        RmHip.Check(native.Ctx, RmHip.rm_blit(native.Ctx, IntPtr.Zero, IntPtr.Zero,
                                             pixels, depth, IntPtr.Zero, w, h));
        frame.LoadRawTextureData(pixels, frameBytes);      // row 0 = bottom row, as Texture2D expects
        frame.Apply(false);
        Graphics.Blit(frame, dest);
    }
}
