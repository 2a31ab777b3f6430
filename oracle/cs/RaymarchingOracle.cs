// RaymarchingOracle.cs -- C# form of the CPU oracle (oracle/rm_oracle.c), Parallel.For over rows.
//
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED BY THE REFERENCE.  Shipped as source: the build
// container has no C# compiler, so every number this repository reports comes from the C
// restatement; this file exists so that a Unity/.NET user can run the same check next to the
// plugin (it needs nothing from UnityEngine).  Same canonical expansions as the C oracle:
// dot = a.x*b.x + a.y*b.y + a.z*b.z left to right, length = sqrt(dot), normalize = v*(1/sqrt(dot)),
// lerp(a,b,t) = a + t*(b-a), min/max drop NaN operands.  All arithmetic is float: each
// product and sum is cast back to float so that a JIT working in double cannot keep excess
// precision.  "cginc" = Assets/Raymarching.cginc, "shader" = Assets/Raymarching.shader.
using System;
using System.Threading.Tasks;

public static class RaymarchingOracle
{
    public struct F3 { public float x, y, z; public F3(float a, float b, float c) { x = a; y = b; z = c; } }
    public struct F4 { public float x, y, z, w; public F4(float a, float b, float c, float d) { x = a; y = b; z = c; w = d; } }

    public sealed class Uniforms
    {
        public float[] FrustumCorners = new float[16], CamInvView = new float[16];       // column-major
        public float[] Positions = new float[16], Rotations = new float[16], Scales = new float[16];
        public F3 CamPos, Light;
        public float MaxSteps = 400, MaxDist = 50, StepFactor = 0.7f, ContactThreshold = 0.002f, NormalSampleScale = 0.00005f;
        public float FarPlane = 1000;
        public F3 Background = new F3(0.19215687f, 0.3019608f, 0.4745098f);
    }

    static float Mul(float a, float b) { return (float)(a * b); }
    static float Add(float a, float b) { return (float)(a + b); }
    static float Sub(float a, float b) { return (float)(a - b); }
    static float Div(float a, float b) { return (float)(a / b); }
    static float Sqrt(float a) { return (float)Math.Sqrt(a); }
    static float Min(float a, float b) { return float.IsNaN(a) ? b : (float.IsNaN(b) ? a : (a < b ? a : b)); }
    static float Max(float a, float b) { return float.IsNaN(a) ? b : (float.IsNaN(b) ? a : (a > b ? a : b)); }
    static float Dot(F3 a, F3 b) { return Add(Add(Mul(a.x, b.x), Mul(a.y, b.y)), Mul(a.z, b.z)); }
    static float Len(F3 a) { return Sqrt(Dot(a, a)); }
    static float Len2(float x, float y) { return Sqrt(Add(Mul(x, x), Mul(y, y))); }
    static float Lerp(float a, float b, float t) { return Add(a, Mul(t, Sub(b, a))); }
    static F3 Lerp(F3 a, F3 b, float t) { return new F3(Lerp(a.x, b.x, t), Lerp(a.y, b.y, t), Lerp(a.z, b.z, t)); }
    static F3 Scale(F3 a, float s) { return new F3(Mul(a.x, s), Mul(a.y, s), Mul(a.z, s)); }
    static F3 AddV(F3 a, F3 b) { return new F3(Add(a.x, b.x), Add(a.y, b.y), Add(a.z, b.z)); }
    static F3 SubV(F3 a, F3 b) { return new F3(Sub(a.x, b.x), Sub(a.y, b.y), Sub(a.z, b.z)); }
    static F3 Normalize(F3 a) { return Scale(a, Div(1f, Sqrt(Dot(a, a)))); }
    static float Clamp(float x, float lo, float hi) { return Min(Max(x, lo), hi); }
    static F4 Row(float[] m, int i) { return new F4(m[i], m[i + 4], m[i + 8], m[i + 12]); }

    // cginc:62-66
    static F3 Rotate(F3 v, F4 q)
    {
        var u = new F3(-q.x, -q.y, -q.z);
        F3 t1 = Scale(u, Mul(2f, Dot(u, v)));
        F3 t2 = Scale(v, Sub(Mul(q.w, q.w), Dot(u, u)));
        var c = new F3(Sub(Mul(u.y, v.z), Mul(u.z, v.y)), Sub(Mul(u.z, v.x), Mul(u.x, v.z)), Sub(Mul(u.x, v.y), Mul(u.y, v.x)));
        return AddV(AddV(t1, t2), Scale(c, Mul(2f, q.w)));
    }
    static float Sphere(F3 p) { return Sub(Len(p), 0.5f); }                                            // cginc:169-171, r = 1
    static float Box(F3 p)                                                                              // cginc:173-177, b = 1
    {
        float dx = Sub(Math.Abs(p.x), 0.5f), dy = Sub(Math.Abs(p.y), 0.5f), dz = Sub(Math.Abs(p.z), 0.5f);
        return Add(Len(new F3(Max(dx, 0f), Max(dy, 0f), Max(dz, 0f))), Min(Max(dx, Max(dy, dz)), 0f));
    }
    static float Link(F3 p, float le, float r1, float r2)                                               // cginc:184-187
    {
        float qy = Max(Sub(Math.Abs(p.y), le), 0f);
        return Sub(Len2(Sub(Len2(p.x, qy), r1), p.z), r2);
    }
    static float Cylinder(F3 p, float h, float r)                                                       // cginc:189-192
    {
        float dx = Sub(Math.Abs(Len2(p.x, p.z)), h), dy = Sub(Math.Abs(p.y), r);
        return Add(Min(Max(dx, dy), 0f), Len2(Max(dx, 0f), Max(dy, 0f)));
    }
    static F4 Smin(F4 a, F4 b, float k)                                                                 // cginc:101-113
    {
        float h = Clamp(Add(0.5f, Div(Mul(0.5f, Sub(b.w, a.w)), k)), 0f, 1f);
        var c1 = new F3(b.x, b.y, b.z); var c2 = new F3(a.x, a.y, a.z);
        F3 col = Scale(Normalize(Lerp(c1, c2, h)), Lerp(Len(c1), Len(c2), h));
        return new F4(col.x, col.y, col.z, Sub(Lerp(b.w, a.w, h), Mul(Mul(k, h), Sub(1f, h))));
    }
    // cginc:334-364
    static F4 Scene(Uniforms u, F3 p)
    {
        var o = new F4[4];
        var col = new[] { new F3(1, .25f, .25f), new F3(.25f, 1, .25f), new F3(.25f, .25f, 1), new F3(1, 1, .25f) };
        for (int i = 0; i < 4; i++)
        {
            F4 P = Row(u.Positions, i), S = Row(u.Scales, i);
            F3 l = Rotate(SubV(p, new F3(P.x, P.y, P.z)), Row(u.Rotations, i));
            l = new F3(Div(l.x, S.x), Div(l.y, S.y), Div(l.z, S.z));
            float d = i == 0 ? Sphere(l) : i == 1 ? Box(l) : i == 2 ? Link(l, 5f, 5f, .1f) : Cylinder(l, .5f, .5f);
            o[i] = new F4(col[i].x, col[i].y, col[i].z, Mul(d, Min(S.x, Min(S.y, S.z))));
        }
        return Smin(Smin(o[0], o[1], .5f), Smin(o[2], o[3], .5f), .5f);
    }

    struct March { public bool hit; public int evals; public float length, contact; }
    // cginc:367-392
    static March Raymarch(Uniforms u, F3 ro, F3 rd, float depth)
    {
        depth = Min(depth, u.MaxDist);
        float steps = Min(2000f, u.MaxSteps), dm = 0f;
        var r = new March { contact = u.ContactThreshold };
        for (int i = 0; (float)i < steps; i++)
        {
            float dts = Scene(u, AddV(ro, Scale(rd, dm))).w;
            r.evals++;
            dm = Add(dm, Mul(Math.Abs(dts), u.StepFactor));
            r.contact = Mul(dm, .0025f);
            if (Math.Abs(dts) < r.contact) { r.hit = true; break; }
            if (dm > depth) break;
        }
        r.length = dm;
        return r;
    }
    // shader:102-123
    static float Shadow(Uniforms u, F3 p, F3 n, float contact, out int evals)
    {
        float depth = Min(2000f, u.MaxDist), steps = Min(2000f, u.MaxSteps), dm = 0f, res = 1f;
        p = AddV(p, Scale(Scale(n, contact), 2f));
        evals = 0;
        for (int i = 0; (float)i < steps; i++)
        {
            float dts = Scene(u, AddV(p, Scale(u.Light, dm))).w;
            evals++;
            res = Min(res, Div(Mul(16f, dts), dm));
            dm = Add(dm, Mul(Math.Abs(dts), u.StepFactor));
            contact = Mul(dm, .0025f);
            if (Math.Abs(dts) < contact) break;
            if (dm > depth) break;
        }
        return res;
    }
    static byte Unorm8(float x) { x = Min(Max(x, 0f), 1f); return (byte)Math.Floor(Add(Mul(x, 255f), 0.5f)); }

    // shader:149-216 for every pixel; row 0 = bottom row.  Returns hit count.
    public static long Render(Uniforms u, int w, int h, byte[] rgba, float[] depthOut, byte[] hitOut)
    {
        F3[] c = new F3[4];
        for (int i = 0; i < 4; i++)
        {
            F4 v = Row(u.FrustumCorners, i); var v3 = new F3(v.x, v.y, v.z);
            F4 r0 = Row(u.CamInvView, 0), r1 = Row(u.CamInvView, 1), r2 = Row(u.CamInvView, 2);
            c[i] = new F3(Dot(new F3(r0.x, r0.y, r0.z), v3), Dot(new F3(r1.x, r1.y, r1.z), v3), Dot(new F3(r2.x, r2.y, r2.z), v3));
        }
        long hits = 0;
        Parallel.For(0, h, () => 0L, (y, _, local) =>
        {
            for (int x = 0; x < w; x++)
            {
                float uu = Div(Add(x, 0.5f), w), vv = Div(Add(y, 0.5f), h);
                F3 rd = Lerp(Lerp(c[3], c[2], uu), Lerp(c[0], c[1], uu), vv);
                March r = Raymarch(u, u.CamPos, rd, u.FarPlane);
                F3 col = u.Background;
                if (r.hit)
                {
                    F3 hp = AddV(u.CamPos, Scale(rd, r.length));
                    float e = Max(0f, u.NormalSampleScale);
                    var raw = new F3(
                        Sub(Scene(u, new F3(Add(hp.x, e), hp.y, hp.z)).w, Scene(u, new F3(Sub(hp.x, e), hp.y, hp.z)).w),
                        Sub(Scene(u, new F3(hp.x, Add(hp.y, e), hp.z)).w, Scene(u, new F3(hp.x, Sub(hp.y, e), hp.z)).w),
                        Sub(Scene(u, new F3(hp.x, hp.y, Add(hp.z, e))).w, Scene(u, new F3(hp.x, hp.y, Sub(hp.z, e))).w));
                    F3 n = Normalize(raw);
                    F4 s = Scene(u, hp);
                    int ev;
                    float shadow = Add(Mul(Div(Sub(Shadow(u, hp, n, r.contact, out ev), 0f), Sub(1f, 0f)), Sub(1f, .25f)), .25f);
                    float ndl = Add(Mul(Dot(n, u.Light), .5f), .5f);
                    float nss = u.NormalSampleScale;
                    float ao = Clamp(Len(new F3(Div(raw.x, nss), Div(raw.y, nss), Div(raw.z, nss))), 0f, 1f);
                    col = Scale(Scale(Scale(new F3(s.x, s.y, s.z), ndl), shadow), ao);
                    local++;
                }
                int idx = y * w + x;
                rgba[4 * idx] = Unorm8(col.x); rgba[4 * idx + 1] = Unorm8(col.y); rgba[4 * idx + 2] = Unorm8(col.z); rgba[4 * idx + 3] = 255;
                depthOut[idx] = r.length;
                hitOut[idx] = (byte)(r.hit ? 1 : 0);
            }
            return local;
        }, local => System.Threading.Interlocked.Add(ref hits, local));
        return hits;
    }
}
