// RaymarchingNative.cs -- plugin-backed counterpart of the project's Raymarching image effect.
// Attach to the camera INSTEAD of Raymarching.cs.  Where that script binds five matrices on
// Raymarching.mat and calls Graphics.Blit(source, destination, mat) (its OnRenderImage), this one
// binds the same names on an rm_ctx and calls rm_blit with the same two inputs the shader reads:
// the source image (_MainTex: what miss pixels show, Raymarching.shader:151,197-202) and the
// camera's depth texture as linear eye depth (Raymarching.shader:154; the script asks for it with
// depthTextureMode = Depth, Raymarching.cs:21), so the skybox stays behind the primitives and meshes
// keep occluding the march.  The marched frame comes back as RGBA8 and is shown with a plain
// texture blit.  The float sliders still live on the Unity material asset and are mirrored into
// the plugin each frame, so the inspector keeps working.
//
// Never compiled (no C# toolchain where this library is built); Unity 2018.3 API names only.
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
    public bool readSource = true;       // pass the source image and the depth texture, as Graphics.Blit(source, ...) does
    public bool pipelined = false;       // rm_blit_async: show frame N - 1 while frame N marches (one frame of latency)
    public Shader linearDepthShader;     // RmLinearEyeDepth.shader (Hidden/RmLinearEyeDepth)

    static readonly string[] FloatNames = {
        "_MaxSteps", "_MaxDist", "_StepFactor", "_ContactThreshold", "_NormalSampleScale",
        "_FractalScale", "_FractalRotationX", "_FractalRotationY", "_FractalRotationZ",
        "_Reflections", "_RefractiveIndex", "_Glossiness" };

    Camera cam;
    RmMaterial native;
    Texture2D frame, srcCopy, depthCopy;
    RenderTexture depthRT;
    Material linearDepthMat;
    // page-locked buffers from rm_host_alloc (the copies run at PCIe rate), two sets of inputs and outputs: a set belongs to
    // the library from rm_blit_async until rm_blit_wait for its ticket has returned
    IntPtr[] pixels = new IntPtr[2], depth = new IntPtr[2], srcPixels = new IntPtr[2], srcDepth = new IntPtr[2];
    long[] tickets = { -1, -1 };
    int frameBytes, frameIndex;

    void OnEnable()
    {
        cam = GetComponent<Camera>();
        cam.depthTextureMode = DepthTextureMode.Depth;          // Raymarching.cs:21
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
            for (int k = 0; k < 2; k++)
            {
                if (tickets[k] >= 0) RmHip.rm_blit_wait(native.Ctx, tickets[k]);      // the library still owns the buffers until then
                foreach (var p in new[] { pixels[k], depth[k], srcPixels[k], srcDepth[k] })
                    if (p != IntPtr.Zero) RmHip.rm_host_free(native.Ctx, p);
            }
        }
        for (int k = 0; k < 2; k++) { pixels[k] = depth[k] = srcPixels[k] = srcDepth[k] = IntPtr.Zero; tickets[k] = -1; }
        if (depthRT != null) { depthRT.Release(); depthRT = null; }
    }

    void Ensure(int w, int h)
    {
        if (frame != null && frame.width == w && frame.height == h) return;
        Release();
        frame = new Texture2D(w, h, TextureFormat.RGBA32, false, true);   // linear: the project is in Gamma space
        srcCopy = new Texture2D(w, h, TextureFormat.RGBA32, false, true);
        depthCopy = new Texture2D(w, h, TextureFormat.RFloat, false, true);
        depthRT = new RenderTexture(w, h, 0, RenderTextureFormat.RFloat, RenderTextureReadWrite.Linear);
        frameBytes = w * h * 4;
        for (int k = 0; k < 2; k++)
        {
            pixels[k] = RmHip.rm_host_alloc(native.Ctx, (UIntPtr)(ulong)frameBytes);
            depth[k] = RmHip.rm_host_alloc(native.Ctx, (UIntPtr)(ulong)frameBytes);
            srcPixels[k] = RmHip.rm_host_alloc(native.Ctx, (UIntPtr)(ulong)frameBytes);
            srcDepth[k] = RmHip.rm_host_alloc(native.Ctx, (UIntPtr)(ulong)frameBytes);
            if (pixels[k] == IntPtr.Zero || depth[k] == IntPtr.Zero || srcPixels[k] == IntPtr.Zero || srcDepth[k] == IntPtr.Zero)
                throw new InvalidOperationException("librmhip: " + RmHip.LastError(native.Ctx));
        }
    }

    // The two textures frag samples, read back into the page-locked buffers of set k: _MainTex as RGBA8 (row 0 = bottom row,
    // the order ReadPixels delivers and rm_blit expects) and the depth texture as linear eye depth,
    // Linear01Depth(tex2D(_CameraDepthTexture, uv)) * _ProjectionParams.z (Raymarching.shader:154), one float per pixel.
    void ReadInputs(RenderTexture src, int k)
    {
        int w = src.width, h = src.height;
        var prev = RenderTexture.active;
        RenderTexture.active = src;
        srcCopy.ReadPixels(new Rect(0, 0, w, h), 0, 0, false);
        byte[] raw = srcCopy.GetRawTextureData();
        Marshal.Copy(raw, 0, srcPixels[k], frameBytes);
        if (linearDepthMat == null)
            linearDepthMat = new Material(linearDepthShader != null ? linearDepthShader : Shader.Find("Hidden/RmLinearEyeDepth"));
        Graphics.Blit(null, depthRT, linearDepthMat);               // the pass reads the global _CameraDepthTexture
        RenderTexture.active = depthRT;
        depthCopy.ReadPixels(new Rect(0, 0, w, h), 0, 0, false);
        raw = depthCopy.GetRawTextureData();
        Marshal.Copy(raw, 0, srcDepth[k], frameBytes);
        RenderTexture.active = prev;
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

        int k = frameIndex & 1;
        IntPtr inPixels = IntPtr.Zero, inDepth = IntPtr.Zero;
        if (readSource)
        {
            if (tickets[k] >= 0) { RmHip.Check(native.Ctx, RmHip.rm_blit_wait(native.Ctx, tickets[k])); tickets[k] = -1; }
            ReadInputs(src, k);
            inPixels = srcPixels[k]; inDepth = srcDepth[k];
        }
        int show = k;
        if (!pipelined)
        {
            RmHip.Check(native.Ctx, RmHip.rm_blit(native.Ctx, inPixels, inDepth, pixels[k], depth[k], IntPtr.Zero, w, h));
        }
        else
        {
            // queue this frame, then wait for the previous one: its copies back ran beside this frame's uploads and march
            if (tickets[k] >= 0) RmHip.Check(native.Ctx, RmHip.rm_blit_wait(native.Ctx, tickets[k]));
            RmHip.Check(native.Ctx, RmHip.rm_blit_async(native.Ctx, inPixels, inDepth, pixels[k], depth[k], IntPtr.Zero, w, h, out tickets[k]));
            if (tickets[k ^ 1] >= 0) { RmHip.Check(native.Ctx, RmHip.rm_blit_wait(native.Ctx, tickets[k ^ 1])); tickets[k ^ 1] = -1; show = k ^ 1; }
            else { RmHip.Check(native.Ctx, RmHip.rm_blit_wait(native.Ctx, tickets[k])); tickets[k] = -1; }      // the very first frame
        }
        frameIndex++;
        frame.LoadRawTextureData(pixels[show], frameBytes);      // row 0 = bottom row, as Texture2D expects
        frame.Apply(false);
        Graphics.Blit(frame, dest);
    }
}
