// RmHip.cs -- P/Invoke surface of librmhip.so (include/rmhip.h), one entry per C symbol.
// Drop this file and librmhip.so (Assets/Plugins/x86_64) into the Unity project.
using System;
using System.Runtime.InteropServices;

public static class RmHip
{
    const string Lib = "rmhip";

    public const int RM_OK = 0, RM_E_ARG = -1, RM_E_DEVICE = -2, RM_E_NOMEM = -3, RM_E_RANGE = -4;
    // Primitive.type and Primitive.op bits (include/rmhip.h)
    public const int RM_SPHERE = 0, RM_BOX = 1, RM_LINK = 2, RM_CYLINDER = 3, RM_TORUS = 4;
    public const int RM_OP_SMAX = 1, RM_OP_NEGATE = 2;
    // values of the _RmSceneBody property
    public const int RM_BODY_PRIMS = 0, RM_BODY_MENGER = 1, RM_BODY_SIERPINSKI = 2, RM_BODY_MANDELBULB = 3, RM_BODY_MANDELBROT = 4, RM_BODY_TERRAIN = 5;
    // rm_set_kernel variants: 0 automatic, 1 tile, 2 persistent queue, 3 lane compaction, 4 tolerance kernel (not bit-exact),
    // 5 four waves per tile that split a primitive table (the automatic choice for tables)
    public const int RM_KERNEL_AUTO = 0, RM_KERNEL_TILE = 1, RM_KERNEL_QUEUE = 2, RM_KERNEL_STREAM = 3, RM_KERNEL_FAST = 4, RM_KERNEL_COOP = 5;

    [StructLayout(LayoutKind.Sequential)]
    public struct Primitive
    {
        public int type, op;
        public float px, py, pz;
        public float qx, qy, qz, qw;
        public float sx, sy, sz;
        public float r, g, b;
        public float p0, p1, p2, p3;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct Stats { public ulong nPixels, nHit, sumStepsPrimary, sumStepsShadow; }

    [DllImport(Lib)] public static extern IntPtr rm_create(int hipDevice);
    [DllImport(Lib)] public static extern void rm_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr rm_last_error(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr rm_version();

    [DllImport(Lib)] public static extern int rm_set_float(IntPtr ctx, string name, float value);
    [DllImport(Lib)] public static extern int rm_set_int(IntPtr ctx, string name, int value);
    [DllImport(Lib)] public static extern int rm_set_vector(IntPtr ctx, string name, float[] v4);
    [DllImport(Lib)] public static extern int rm_set_matrix(IntPtr ctx, string name, float[] m16);
    [DllImport(Lib)] public static extern int rm_get_float(IntPtr ctx, string name, out float value);
    [DllImport(Lib)] public static extern int rm_get_int(IntPtr ctx, string name, out int value);
    [DllImport(Lib)] public static extern int rm_get_vector(IntPtr ctx, string name, [Out] float[] v4);
    [DllImport(Lib)] public static extern int rm_get_matrix(IntPtr ctx, string name, [Out] float[] m16);
    [DllImport(Lib)] public static extern int rm_has_property(IntPtr ctx, string name);
    [DllImport(Lib)] public static extern int rm_pack_rgb_device(IntPtr ctx, IntPtr dRgba8, IntPtr dRgb8, UIntPtr nPixels, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_unpack_rgb_device(IntPtr ctx, IntPtr dRgb8, IntPtr dRgba8, UIntPtr nPixels, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_set_primitives(IntPtr ctx, Primitive[] prims, int n);

    [DllImport(Lib)] public static extern int rm_blit(IntPtr ctx, IntPtr srcRgba8, IntPtr sceneDepth,
                                                      IntPtr dstRgba8, IntPtr dstDepth, IntPtr dstHit, int w, int h);
    // rm_blit in two halves: queue frame N + 1, then wait for frame N (its copies back overlap the next march)
    [DllImport(Lib)] public static extern int rm_blit_async(IntPtr ctx, IntPtr srcRgba8, IntPtr sceneDepth,
                                                            IntPtr dstRgba8, IntPtr dstDepth, IntPtr dstHit, int w, int h, out long ticket);
    [DllImport(Lib)] public static extern int rm_blit_wait(IntPtr ctx, long ticket);
    [DllImport(Lib)] public static extern int rm_blit_device(IntPtr ctx, IntPtr dSrc, IntPtr dSceneDepth,
                                                             IntPtr dRgba8, IntPtr dDepth, IntPtr dHit, int w, int h, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_blit_device_noalloc(IntPtr ctx, IntPtr dSrc, IntPtr dSceneDepth,
                                                                     IntPtr dRgba8, IntPtr dDepth, IntPtr dHit, int w, int h, IntPtr hipStream);
    // dstSteps: two ushort per pixel (primary, shadow evaluations)
    [DllImport(Lib)] public static extern int rm_blit_steps(IntPtr ctx, IntPtr srcRgba8, IntPtr sceneDepth,
                                                            IntPtr dstRgba8, IntPtr dstDepth, IntPtr dstHit, IntPtr dstSteps, int w, int h);
    [DllImport(Lib)] public static extern int rm_blit_steps_device(IntPtr ctx, IntPtr dSrc, IntPtr dSceneDepth, IntPtr dRgba8, IntPtr dDepth,
                                                                   IntPtr dHit, IntPtr dSteps, int w, int h, IntPtr hipStream);
    // page-locked host memory for rm_blit's images (copies at PCIe rate instead of staged through pageable memory)
    [DllImport(Lib)] public static extern IntPtr rm_host_alloc(IntPtr ctx, UIntPtr bytes);
    [DllImport(Lib)] public static extern void rm_host_free(IntPtr ctx, IntPtr p);
    // Material.SetTexture("_ShadowMap", ...) for a depth plane in HBM (w x h floats)
    [DllImport(Lib)] public static extern int rm_set_texture_device(IntPtr ctx, string name, IntPtr dTexels, int w, int h);
    // compact frame transport between GPUs: records of rm_tile_record_bytes() per 8x8 tile that holds a hit
    [DllImport(Lib)] public static extern UIntPtr rm_tile_record_bytes();
    [DllImport(Lib)] public static extern int rm_compact_frames_device(IntPtr ctx, IntPtr dRgba8, IntPtr dHit, int w, int h, int nFrames,
                                                                       IntPtr dRecords, IntPtr dCount, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_expand_frames_device(IntPtr ctx, IntPtr dRecords, uint nRecords, IntPtr dSrcRgba8, IntPtr dDstRgba8,
                                                                      int w, int h, int nFrames, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_update_frames_device(IntPtr ctx, IntPtr dOldRecords, uint nOld, IntPtr dNewRecords, uint nNew,
                                                                      IntPtr dSrcRgba8, IntPtr dDstRgba8, int w, int h, int nFrames, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_band_rows(int h, int bandRows, int part, int nParts);
    [DllImport(Lib)] public static extern int rm_blit_bands_device(IntPtr ctx, IntPtr dSrc, IntPtr dSceneDepth, IntPtr dRgba8, IntPtr dDepth,
                                                                   IntPtr dHit, int w, int h, int bandRows, int part, int nParts, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_blit_poses_device(IntPtr ctx, int nPoses, float[] frustumCorners, float[] camInvView, float[] camPos,
                                                                   IntPtr dRgba8, IntPtr dDepth, IntPtr dHit, int w, int h, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_shadowmap_render(IntPtr ctx, IntPtr dstDepth, int w, int h);
    [DllImport(Lib)] public static extern int rm_shadowmap_render_device(IntPtr ctx, IntPtr dDepth, IntPtr dHit, int w, int h, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_shadowmap_directional(IntPtr ctx, float[] corners12, float[] forward3, IntPtr dstDepth, int w, int h);
    [DllImport(Lib)] public static extern int rm_shadowmap_directional_device(IntPtr ctx, float[] corners12, float[] forward3, IntPtr dDepth, IntPtr dHit,
                                                                              int w, int h, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_shadowmap_directional_bands_device(IntPtr ctx, float[] corners12, float[] forward3, IntPtr dDepth, IntPtr dHit,
                                                                                    int w, int h, int bandRows, int part, int nParts, IntPtr hipStream);
    [DllImport(Lib)] public static extern int rm_get_stats(IntPtr ctx, out Stats stats);
    [DllImport(Lib)] public static extern int rm_get_leaf_stats(IntPtr ctx, out ulong visited, out ulong possible);
    [DllImport(Lib)] public static extern int rm_debug_counter_slots(IntPtr ctx, [Out] uint[] words, int maxWords, out int wordsPerSlot, out int nSlots);
    [DllImport(Lib)] public static extern int rm_set_kernel(IntPtr ctx, int variant);

    public static string LastError(IntPtr ctx) { return Marshal.PtrToStringAnsi(rm_last_error(ctx)); }

    public static void Check(IntPtr ctx, int rc)
    {
        if (rc != RM_OK) throw new InvalidOperationException("librmhip " + rc + ": " + LastError(ctx));
    }

    // UnityEngine.Matrix4x4 is sixteen floats in column-major field order: it can be passed as is.
    public static float[] Floats(UnityEngine.Matrix4x4 m)
    {
        return new[] { m.m00, m.m10, m.m20, m.m30, m.m01, m.m11, m.m21, m.m31, m.m02, m.m12, m.m22, m.m32, m.m03, m.m13, m.m23, m.m33 };
    }
}

// The property-bag half of UnityEngine.Material, backed by an rm_ctx.
public sealed class RmMaterial : IDisposable
{
    public IntPtr Ctx { get; private set; }

    public RmMaterial(int hipDevice = 0)
    {
        Ctx = RmHip.rm_create(hipDevice);
        if (Ctx == IntPtr.Zero) throw new InvalidOperationException("rm_create: " + RmHip.LastError(IntPtr.Zero));
    }

    public void SetFloat(string name, float v) { RmHip.Check(Ctx, RmHip.rm_set_float(Ctx, name, v)); }
    public void SetInt(string name, int v) { RmHip.Check(Ctx, RmHip.rm_set_int(Ctx, name, v)); }
    public void SetVector(string name, UnityEngine.Vector4 v) { RmHip.Check(Ctx, RmHip.rm_set_vector(Ctx, name, new[] { v.x, v.y, v.z, v.w })); }
    public void SetMatrix(string name, UnityEngine.Matrix4x4 m) { RmHip.Check(Ctx, RmHip.rm_set_matrix(Ctx, name, RmHip.Floats(m))); }
    public float GetFloat(string name) { float v; RmHip.Check(Ctx, RmHip.rm_get_float(Ctx, name, out v)); return v; }
    public bool HasProperty(string name) { return RmHip.rm_has_property(Ctx, name) != 0; }

    // copies the floats the inspector serialized on the Unity material into the plugin
    public void CopyFloatsFrom(UnityEngine.Material mat, params string[] names)
    {
        foreach (var n in names) if (mat.HasProperty(n)) SetFloat(n, mat.GetFloat(n));
    }

    public void Dispose()
    {
        if (Ctx != IntPtr.Zero) { RmHip.rm_destroy(Ctx); Ctx = IntPtr.Zero; }
    }
}
