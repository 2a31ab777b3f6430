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
//
// Covered: the 4-slot scene of the three matrices, N-primitive tables (five primitive types, smax / negation bits, the
// level-wise pairwise tree), the other scene() bodies (menger, sierpinski, mandelbulb, mandelbrot, terrain and the _Time-driven
// composite of cginc:349) with the canonical double-precision series of their transcendental functions, the domain operators,
// _MainTex and the depth texture as inputs, the shadow-map lookup in both modes, the two depth passes, per-pixel step counts.
// Not covered (C oracle only): frag's optional lines (bounce, debug views, step term, light integration, light models).
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
        public float SmoothK = 0.5f;
        public Primitive[] Table;                 // null: the 4-slot scene of Positions / Rotations / Scales
        // shadow-map lookup in place of the shadow march (0 off, 1 perspective map through ShadowVP, 2 directional map)
        public int ShadowMode;
        public float[] ShadowVP = new float[16];  // projection * worldToCamera of the shadow camera, column-major
        public float[] ShadowCorners = new float[12]; public F3 ShadowForward;     // rectangle TL, TR, BR, BL and its forward
        public float[] ShadowMap; public int ShadowW, ShadowH; public float ShadowBias = 0.01f;
        // scene() bodies other than the primitives (0): 1 menger, 2 sierpinski, 3 mandelbulb, 4 mandelbrot, 5 terrain, 6 composite
        public int SceneBody;
        public float[] FractalRotation = new float[3];          // _FractalRotationX/Y/Z
        public float FractalScale = 2f, FractalPower = 0f;       // power 0: 7 for the mandelbulb (cginc:194), 2 for the mandelbrot (:229)
        public float TimeY;                                      // _Time.y
        public bool AnimateFractal;                              // cginc:346-348
        public int DomainOp; public F3 DomainPeriod; public float DomainSize;   // 1 shmod repetition, 2 boxfold, 3 spherefold
    }

    public const int Sphere_ = 0, Box_ = 1, Link_ = 2, Cylinder_ = 3, Torus_ = 4, OpSmax = 1, OpNegate = 2;
    // one row of a primitive table: the fields of RmHip.Primitive
    public struct Primitive
    {
        public int type, op;
        public F3 pos; public F4 quat; public F3 scale, color;
        public float p0, p1, p2, p3;
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
    static float Sphere(F3 p, float r) { return Sub(Len(p), Mul(r, 0.5f)); }                            // cginc:169-171
    static float Box(F3 p, F3 b)                                                                        // cginc:173-177
    {
        float dx = Sub(Math.Abs(p.x), Mul(b.x, 0.5f)), dy = Sub(Math.Abs(p.y), Mul(b.y, 0.5f)), dz = Sub(Math.Abs(p.z), Mul(b.z, 0.5f));
        return Add(Len(new F3(Max(dx, 0f), Max(dy, 0f), Max(dz, 0f))), Min(Max(dx, Max(dy, dz)), 0f));
    }
    static float Torus(F3 p, float tx, float ty) { return Sub(Len2(Sub(Len2(p.x, p.y), tx), p.z), ty); } // cginc:179-182
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
    // cginc:120-123: plain lerp of rgb, and the reference's sign of the k*h*(1-h) term
    static F4 Smax(F4 a, F4 b, float k)
    {
        float h = Clamp(Add(0.5f, Div(Mul(0.5f, Sub(a.w, b.w)), k)), 0f, 1f);
        F3 col = Lerp(new F3(b.x, b.y, b.z), new F3(a.x, a.y, a.z), h);
        return new F4(col.x, col.y, col.z, Sub(-Lerp(-b.w, -a.w, h), Mul(Mul(k, h), Sub(1f, h))));
    }
    // the 4-slot scene as a table: sphere, box, link(5,5,.1), cylinder(.5,.5) with the colours of cginc:351-359
    static Primitive[] LegacyTable(Uniforms u)
    {
        var col = new[] { new F3(1, .25f, .25f), new F3(.25f, 1, .25f), new F3(.25f, .25f, 1), new F3(1, 1, .25f) };
        var par = new[] { new F4(1, 0, 0, 0), new F4(1, 1, 1, 0), new F4(5, 5, .1f, 0), new F4(.5f, .5f, 0, 0) };
        var t = new Primitive[4];
        for (int i = 0; i < 4; i++)
        {
            F4 P = Row(u.Positions, i), S = Row(u.Scales, i);
            t[i] = new Primitive { type = i, op = 0, pos = new F3(P.x, P.y, P.z), quat = Row(u.Rotations, i), scale = new F3(S.x, S.y, S.z),
                                   color = col[i], p0 = par[i].x, p1 = par[i].y, p2 = par[i].z, p3 = par[i].w };
        }
        return t;
    }
    // the wrapper of cginc:336-354 around one primitive: prim(rotate(p - pos, q) / scale) * min(scale), negated on request
    static float PrimDist(Primitive pr, F3 p)
    {
        F3 l = Rotate(SubV(p, pr.pos), pr.quat);
        l = new F3(Div(l.x, pr.scale.x), Div(l.y, pr.scale.y), Div(l.z, pr.scale.z));
        float d;
        switch (pr.type)
        {
            case Sphere_: d = Sphere(l, pr.p0); break;
            case Box_: d = Box(l, new F3(pr.p0, pr.p1, pr.p2)); break;
            case Link_: d = Link(l, pr.p0, pr.p1, pr.p2); break;
            case Cylinder_: d = Cylinder(l, pr.p0, pr.p1); break;
            default: d = Torus(l, pr.p0, pr.p1); break;
        }
        d = Mul(d, Min(pr.scale.x, Min(pr.scale.y, pr.scale.z)));
        return (pr.op & OpNegate) != 0 ? -d : d;
    }
    // cginc:334-364.  Four primitives: smin(smin(o0,o1), smin(o2,o3)) (cginc:363).  Any other count: the same pairwise tree,
    // level by level in index order, an odd element at the end of a level carried up unchanged; the op bits of the first leaf
    // of a node's right operand choose smax for that node.
    static F4 Scene(Ctx c, F3 p)
    {
        Uniforms u = c.u; Primitive[] t = c.t;
        p = Domain(u, p);
        if (u.SceneBody != 0) return SceneFractal(c, p);
        int n = t.Length;
        var o = new F4[n]; var first = new int[n];
        for (int i = 0; i < n; i++) { o[i] = new F4(t[i].color.x, t[i].color.y, t[i].color.z, PrimDist(t[i], p)); first[i] = i; }
        while (n > 1)
        {
            int m = 0;
            for (int i = 0; i + 1 < n; i += 2)
            {
                bool smax = (t[first[i + 1]].op & OpSmax) != 0;
                first[m] = first[i];
                o[m++] = smax ? Smax(o[i], o[i + 1], u.SmoothK) : Smin(o[i], o[i + 1], u.SmoothK);
            }
            if ((n & 1) != 0) { first[m] = first[n - 1]; o[m++] = o[n - 1]; }
            n = m;
        }
        return o[0];
    }

    // ---- cginc:51-60 rotate(v, euler angles) = mul(mz, mul(my, mul(mx, v))): cos / sin of the uniform angles once per frame as
    // the double-precision library value rounded to float (Math.Cos / Math.Sin / Math.Pow of this runtime: a library whose double
    // result differs from the C library's in the last bit changes the float only on a rounding boundary); the constant 0 / 1
    // matrix entries folded
    struct Euler { public float cx, sx, cy, sy, cz, sz; }
    static Euler EulerOf(float ax, float ay, float az)
    {
        return new Euler { cx = (float)Math.Cos(ax), sx = (float)Math.Sin(ax), cy = (float)Math.Cos(ay), sy = (float)Math.Sin(ay),
                           cz = (float)Math.Cos(az), sz = (float)Math.Sin(az) };
    }
    static F3 RotateEuler(F3 v, Euler e)
    {
        var a = new F3(v.x, Add(Mul(e.cx, v.y), Mul(-e.sx, v.z)), Add(Mul(e.sx, v.y), Mul(e.cx, v.z)));
        var b = new F3(Add(Mul(e.cy, a.x), Mul(e.sy, a.z)), a.y, Add(Mul(-e.sy, a.x), Mul(e.cy, a.z)));
        return new F3(Add(Mul(e.cz, b.x), Mul(-e.sz, b.y)), Add(Mul(e.sz, b.x), Mul(e.cz, b.y)), b.z);
    }

    // ---- canonical transcendental functions: float -> float, evaluated in IEEE double with + - * / sqrt in exactly this order,
    // rounded to float once (HLSL leaves their precision to the driver); the same definitions as in oracle/rm_oracle.c
    const double PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11, PI = 3.14159265358979311600e+00;
    const double PIO2 = 1.57079632679489655800e+00, PIO4 = 7.85398163397448278999e-01;
    const double LN2 = 6.93147180559945286227e-01, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    static double SinPoly(double r)
    {
        double q = r * r, p = 1.0 / 355687428096000.0;
        p = p * q - 1.0 / 1307674368000.0; p = p * q + 1.0 / 6227020800.0; p = p * q - 1.0 / 39916800.0; p = p * q + 1.0 / 362880.0;
        p = p * q - 1.0 / 5040.0; p = p * q + 1.0 / 120.0; p = p * q - 1.0 / 6.0;
        return r + r * (q * p);
    }
    static double CosPoly(double r)
    {
        double q = r * r, p = 1.0 / 20922789888000.0;
        p = p * q - 1.0 / 87178291200.0; p = p * q + 1.0 / 479001600.0; p = p * q - 1.0 / 3628800.0; p = p * q + 1.0 / 40320.0;
        p = p * q - 1.0 / 720.0; p = p * q + 1.0 / 24.0; p = p * q - 0.5;
        return 1.0 + q * p;
    }
    static double Reduce(double x, out int quadrant)
    {
        double k = Math.Floor(x * (1.0 / PIO2) + 0.5);
        double r = (x - k * PIO2_HI) - k * PIO2_LO;
        quadrant = (int)(k - 4.0 * Math.Floor(k * 0.25));
        return r;
    }
    static float CSin(float xf) { int n; double r = Reduce(xf, out n); double v = (n & 1) != 0 ? CosPoly(r) : SinPoly(r); return (float)((n & 2) != 0 ? -v : v); }
    static float CCos(float xf) { int n; double r = Reduce(xf, out n); double v = (n & 1) != 0 ? SinPoly(r) : CosPoly(r); return (float)(((n + 1) & 2) != 0 ? -v : v); }
    static double Atan01(double t)
    {
        double b = 0.0, u = t;
        if (t > 0.41421356237309503) { u = (t - 1.0) / (t + 1.0); b = PIO4; }
        double q = u * u, p = 1.0 / 43.0;
        for (int n = 41; n >= 1; n -= 2) p = 1.0 / (double)n - q * p;
        return b + u * p;
    }
    static double Atan2d(double y, double x)
    {
        double ay = Math.Abs(y), ax = Math.Abs(x), a;
        if (ay > ax) a = PIO2 - Atan01(ax / ay);
        else if (ax > 0.0) a = Atan01(ay / ax);
        else a = ay / ax;                                   // 0/0 and NaN stay NaN
        if (x < 0.0) a = PI - a;
        return y < 0.0 ? -a : a;
    }
    static float CAtan2(float y, float x) { return (float)Atan2d(y, x); }
    static float CAcos(float vf) { double v = vf, t = 1.0 - v * v; if (!(t > 0.0)) t = 0.0; return (float)Atan2d(Math.Sqrt(t), v); }
    static float CAsin(float vf) { double v = vf, t = 1.0 - v * v; if (!(t > 0.0)) t = 0.0; return (float)Atan2d(v, Math.Sqrt(t)); }
    static double Logd(double x)                            // finite x > 0 that is a normal double (every positive float is)
    {
        long bits = BitConverter.DoubleToInt64Bits(x);
        int e = (int)((bits >> 52) & 0x7ff) - 1022;         // x = m * 2^e with m in [0.5, 1): frexp
        double m = BitConverter.Int64BitsToDouble((bits & ~(0x7ffL << 52)) | (1022L << 52)) * 2.0;
        e -= 1;
        if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
        double s = (m - 1.0) / (m + 1.0), q = s * s, p = 1.0 / 25.0;
        for (int n = 23; n >= 1; n -= 2) p = 1.0 / (double)n + q * p;
        return (double)e * LN2 + 2.0 * s * p;
    }
    static readonly double[] InvFact = { 1.0, 1.0, 1.0 / 2.0, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0,
                                         1.0 / 362880.0, 1.0 / 3628800.0, 1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0 };
    static double Expd(double y)
    {
        if (!(y < 700.0)) return y > 0.0 ? double.PositiveInfinity : y;
        if (y < -700.0) return 0.0;
        double k = Math.Floor(y * (1.0 / LN2) + 0.5);
        double f = (y - k * LN2_HI) - k * LN2_LO, p = 1.0 / 87178291200.0;
        for (int n = 13; n >= 0; n--) p = InvFact[n] + f * p;
        return p * BitConverter.Int64BitsToDouble((long)((int)k + 1023) << 52);      // ldexp(p, k), |k| <= 1010: exact
    }
    static float CLog(float xf)
    {
        if (!(xf > 0f)) return xf == 0f ? float.NegativeInfinity : float.NaN;
        if (float.IsInfinity(xf)) return xf;
        return (float)Logd(xf);
    }
    static float CPow(float xf, float ef)                   // x >= 0 only (a length)
    {
        if (!(xf > 0f)) { if (xf == 0f) return ef > 0f ? 0f : (ef == 0f ? 1f : float.PositiveInfinity); return float.NaN; }
        if (float.IsInfinity(xf)) return ef > 0f ? xf : (ef == 0f ? 1f : 0f);
        return (float)Expd((double)ef * Logd(xf));
    }

    // cginc:194-227 (brot false) and cginc:229-262 (brot true); iters = 12, bailout = 10
    static F4 Bulb(F3 p, float e, bool brot)
    {
        F3 z = p;
        float dr = 1f, r = 0f, o = 10f, o2 = 10f, o3 = 10f;
        for (int i = 0; i < 12; i++)
        {
            if (!brot) r = Len(z);
            o = Min(o, Len(z));
            o2 = Min(o2, brot ? Len(new F3(Sub(z.x, 1f), z.y, z.z)) : Len(new F3(z.x, z.y, Sub(z.z, 1f))));
            o3 = Min(o3, brot ? Len(new F3(Sub(z.x, 2f), z.y, z.z)) : Len(new F3(z.x, z.y, Sub(z.z, 2f))));
            if (brot) r = Len(z);
            if (r > 10f) break;
            float theta = brot ? CAtan2(z.y, z.x) : CAcos(Div(z.z, r));
            float phi = brot ? CAsin(Div(z.z, r)) : CAtan2(z.y, z.x);
            dr = Add(Mul(Mul(CPow(r, Sub(e, 1f)), e), dr), 1f);
            float zr = CPow(r, e);
            theta = Mul(theta, e);
            phi = Mul(phi, e);
            if (brot) z = new F3(Mul(zr, Mul(CCos(theta), CCos(phi))), Mul(zr, Mul(CSin(theta), CCos(phi))), Mul(zr, -CSin(phi)));
            else z = new F3(Mul(zr, Mul(CSin(theta), CCos(phi))), Mul(zr, Mul(CSin(phi), CSin(theta))), Mul(zr, CCos(theta)));
            z = AddV(z, p);
        }
        return new F4(o, o2, o3, Div(Mul(Mul(0.5f, CLog(r)), r), dr));
    }
    // cginc:295-313: rgb = orbit traps, w = distance estimate; pow3 = pow(3.0, -12)
    static F4 Menger(F3 p, Euler rot, float pow3)
    {
        float x = p.x, y = p.y, z = p.z, t, o = 50f, o2 = 50f, o3 = 50f;
        for (int n = 0; n < 12; n++)
        {
            x = Math.Abs(x); y = Math.Abs(y); z = Math.Abs(z);
            if (x < y) { t = x; x = y; y = t; }
            if (y < z) { t = y; y = z; z = t; }
            if (x < y) { t = x; x = y; y = t; }
            F3 q = RotateEuler(new F3(x, y, z), rot);
            x = Sub(Mul(q.x, 3f), 2f); y = Sub(Mul(q.y, 3f), 2f); z = Sub(Mul(q.z, 3f), 2f);
            if (z < -1f) z = Add(z, 2f);
            o = Min(o, Len(new F3(x, y, z)));
            o2 = Min(o2, Len(new F3(x, Sub(y, 0.5f), z)));
            o3 = Min(o3, Len(new F3(x, y, Sub(z, 0.5f))));
        }
        return new F4(o, o2, o3, Mul(Sub(Sqrt(Add(Add(Mul(x, x), Mul(y, y)), Mul(z, z))), 1.5f), pow3));
    }
    // cginc:264-293; pows[i] = pow(scale, -i) for the 16 values the loop counter can end on
    static F4 Sierpinski(F3 p, float scale, Euler rot, float[] pows)
    {
        float x = p.x, y = p.y, z = p.z, r = Add(Add(Mul(x, x), Mul(y, y)), Mul(z, z)), o = 20f, o2 = 20f, o3 = 20f;
        int i;
        for (i = 0; i < 15 && r < 20f; i++)
        {
            if (Add(x, y) < 0f) { float x1 = -y; y = -x; x = x1; }
            if (Add(x, z) < 0f) { float x1 = -z; z = -x; x = x1; }
            if (Add(y, z) < 0f) { float y1 = -z; z = -y; y = y1; }
            F3 c = RotateEuler(new F3(x, y, z), rot);
            float s1 = Sub(scale, 1f);
            x = Sub(Mul(scale, c.x), s1); y = Sub(Mul(scale, c.y), s1); z = Sub(Mul(scale, c.z), s1);
            r = Add(Add(Mul(x, x), Mul(y, y)), Mul(z, z));
            o = Min(o, Len(new F3(Sub(x, 1f), y, z)));
            o2 = Min(o2, Len(new F3(x, Sub(y, 1f), z)));
            o3 = Min(o3, Len(new F3(x, y, Sub(z, 1f))));
        }
        return new F4(o, o2, o3, Mul(Sub(Sqrt(r), 2f), pows[i]));
    }
    // cginc:125-149: rand / noise with the canonical sin and frac(x) = x - floor(x)
    static float Frac(float x) { return Sub(x, (float)Math.Floor(x)); }
    static float Rand2(float x, float y) { return Frac(Mul(CSin(Add(Mul(x, 12.9898f), Mul(y, 78.233f))), 43758.5453f)); }
    static float Noise2(float x, float y)
    {
        float ix = (float)Math.Floor(x), iy = (float)Math.Floor(y), fx = Frac(x), fy = Frac(y);
        float a = Rand2(ix, iy), b = Rand2(Add(ix, 1f), Add(iy, 0f)), c = Rand2(Add(ix, 0f), Add(iy, 1f)), d = Rand2(Add(ix, 1f), Add(iy, 1f));
        float ux = Mul(Mul(fx, fx), Sub(3f, Mul(2f, fx))), uy = Mul(Mul(fy, fy), Sub(3f, Mul(2f, fy)));
        return Add(Add(Lerp(a, b, ux), Mul(Mul(Sub(c, a), uy), Sub(1f, ux))), Mul(Mul(Sub(d, b), ux), uy));
    }
    // cginc:315-330, terrain(x).w: `float2 n = noise(p)` broadcasts the scalar, d accumulates in .x only
    static float TerrainW(float x, float y)
    {
        float px = Div(Mul(x, 0.003f), 250f), py = Div(Mul(y, 0.003f), 250f), a = 0f, b = 1f, dx = 0f, dy = 0f;
        for (int i = 0; i < 3; i++)
        {
            float n = Noise2(px, py);
            dx = Add(dx, n); dy = Add(dy, 0f);
            a = Add(a, Div(Mul(b, n), Add(1f, Add(Mul(dx, dx), Mul(dy, dy)))));
            b = Mul(b, 0.5f);
            px = Mul(Mul(.8f, px), 2f); py = Mul(Mul(-.6f, py), 2f);
        }
        return Mul(Mul(250f, 120f), a);
    }
    // cginc:84-99 mod / shmod (HLSL fmod taken as the exact remainder, which C#'s % is), cginc:156-165 boxfold / spherefold
    static float Mod1(float x, float m) { float r = x % m; return r < 0f ? Add(r, m) : r; }
    static float Shmod1(float x, float m) { return Sub(Mod1(Add(x, Mul(m, .5f)), m), Mul(m, .5f)); }
    static F3 Domain(Uniforms u, F3 p)
    {
        switch (u.DomainOp)
        {
            case 1:
                if (u.DomainPeriod.x > 0f) p.x = Shmod1(p.x, u.DomainPeriod.x);
                if (u.DomainPeriod.y > 0f) p.y = Shmod1(p.y, u.DomainPeriod.y);
                if (u.DomainPeriod.z > 0f) p.z = Shmod1(p.z, u.DomainPeriod.z);
                break;
            case 2:
            {
                float b = Mul(u.DomainSize, .5f);
                p = new F3(Sub(Mul(Clamp(p.x, -b, b), 2f), p.x), Sub(Mul(Clamp(p.y, -b, b), 2f), p.y), Sub(Mul(Clamp(p.z, -b, b), 2f), p.z));
                break;
            }
            case 3:
            {
                float R = u.DomainSize, r = Len(p);
                if (r < R) { float rr = Mul(r, r); p = new F3(Div(Mul(Mul(p.x, R), R), rr), Div(Mul(Mul(p.y, R), R), rr), Div(Mul(Mul(p.z, R), R), rr)); }
                break;
            }
        }
        return p;
    }

    // everything of a frame that does not depend on the pixel
    sealed class Ctx
    {
        public Uniforms u; public Primitive[] t; public Primitive slot0;
        public Euler frot, rotTime, rotMenger; public float pow3, fpower; public float[] pows = new float[16];
        public Ctx(Uniforms u_)
        {
            u = u_;
            Primitive[] legacy = LegacyTable(u);
            t = u.Table ?? legacy; slot0 = legacy[0];
            float ty = u.TimeY;
            frot = u.AnimateFractal                                                            // cginc:346-348
                ? EulerOf(Mul((float)Math.Sin(ty), .5f), Mul((float)Math.Sin(Mul(ty, 1.414f)), .5f), Mul((float)Math.Sin(Mul(ty, 1.618f)), .5f))
                : EulerOf(u.FractalRotation[0], u.FractalRotation[1], u.FractalRotation[2]);
            rotTime = EulerOf(0f, 0f, ty);                                                     // cginc:349
            rotMenger = EulerOf(Mul(ty, .25f), Mul(Mul(ty, 1.4f), .25f), Mul(Mul(ty, 1.6f), .25f));
            pow3 = (float)Math.Pow(3.0, -12.0);
            fpower = u.FractalPower > 0f ? u.FractalPower : (u.SceneBody == 3 ? 7f : 2f);
            for (int i = 0; i < 16; i++) pows[i] = (float)Math.Pow(u.FractalScale, -(double)i);
        }
    }
    // the bodies other than the primitives: the wrapper of cginc:336-354 with slot 0 around a fractal (body 6: the world-space p)
    static F4 SceneFractal(Ctx c, F3 p)
    {
        if (c.u.SceneBody == 6)                                                                // cginc:349
            return Bulb(RotateEuler(p, c.rotTime), Add(Menger(RotateEuler(p, c.rotMenger), c.frot, c.pow3).w, 1.1f), false);
        Primitive pr = c.slot0;
        F3 l = Rotate(SubV(p, pr.pos), pr.quat);
        l = new F3(Div(l.x, pr.scale.x), Div(l.y, pr.scale.y), Div(l.z, pr.scale.z));
        F4 o;
        switch (c.u.SceneBody)
        {
            case 1: o = Menger(l, c.frot, c.pow3); break;
            case 2: o = Sierpinski(l, c.u.FractalScale, c.frot, c.pows); break;
            case 3: o = Bulb(l, c.fpower, false); break;
            case 5: o = new F4(1f, 1f, 1f, Sub(l.y, TerrainW(l.x, l.z))); break;
            default: o = Bulb(l, c.fpower, true); break;
        }
        o.w = Mul(o.w, Min(pr.scale.x, Min(pr.scale.y, pr.scale.z)));
        return o;
    }

    struct March { public bool hit; public int evals; public float length, contact; }
    // cginc:367-392
    static March Raymarch(Ctx cx, F3 ro, F3 rd, float depth)
    {
        Uniforms u = cx.u;
        depth = Min(depth, u.MaxDist);
        float steps = Min(2000f, u.MaxSteps), dm = 0f;
        var r = new March { contact = u.ContactThreshold };
        for (int i = 0; (float)i < steps; i++)
        {
            float dts = Scene(cx, AddV(ro, Scale(rd, dm))).w;
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
    static float Shadow(Ctx cx, F3 p, F3 n, float contact, out int evals)
    {
        Uniforms u = cx.u;
        float depth = Min(2000f, u.MaxDist), steps = Min(2000f, u.MaxSteps), dm = 0f, res = 1f;
        p = AddV(p, Scale(Scale(n, contact), 2f));
        evals = 0;
        for (int i = 0; (float)i < steps; i++)
        {
            float dts = Scene(cx, AddV(p, Scale(u.Light, dm))).w;
            evals++;
            res = Min(res, Div(Mul(16f, dts), dm));
            dm = Add(dm, Mul(Math.Abs(dts), u.StepFactor));
            contact = Mul(dm, .0025f);
            if (Math.Abs(dts) < contact) break;
            if (dm > depth) break;
        }
        return res;
    }
    // Shadow-map lookup in place of the shadow march (this library's definition; the reference declares _ShadowMap and
    // _ShadowMapVPMatrix at shader:45-46 and never reads them): 1 = lit, 0 = in shadow; lit outside the map.
    static float ShadowLookup(Uniforms u, F3 p, F3 n, float contact)
    {
        F3 q = AddV(p, Scale(Scale(n, contact), 2f));                                                 // shader:107
        float uu, vv, d;
        if (u.ShadowMode == 1)
        {
            F4 r0 = Row(u.ShadowVP, 0), r1 = Row(u.ShadowVP, 1), r3 = Row(u.ShadowVP, 3);
            float cx = Add(Add(Add(Mul(r0.x, q.x), Mul(r0.y, q.y)), Mul(r0.z, q.z)), r0.w);
            float cy = Add(Add(Add(Mul(r1.x, q.x), Mul(r1.y, q.y)), Mul(r1.z, q.z)), r1.w);
            float cw = Add(Add(Add(Mul(r3.x, q.x), Mul(r3.y, q.y)), Mul(r3.z, q.z)), r3.w);
            if (!(cw > 0f)) return 1f;
            uu = Add(Mul(Div(cx, cw), .5f), .5f);
            vv = Add(Mul(Div(cy, cw), .5f), .5f);
            d = cw;
        }
        else
        {
            float[] k = u.ShadowCorners;
            var bl = new F3(k[9], k[10], k[11]);
            F3 e1 = SubV(new F3(k[6], k[7], k[8]), bl), e2 = SubV(new F3(k[0], k[1], k[2]), bl), r = SubV(q, bl);
            uu = Div(Dot(r, e1), Dot(e1, e1));
            vv = Div(Dot(r, e2), Dot(e2, e2));
            d = Dot(r, u.ShadowForward);
        }
        if (!(uu >= 0f && uu < 1f && vv >= 0f && vv < 1f)) return 1f;
        int ix = (int)Math.Floor(Mul(uu, u.ShadowW)), iy = (int)Math.Floor(Mul(vv, u.ShadowH));
        if (ix > u.ShadowW - 1) ix = u.ShadowW - 1;
        if (iy > u.ShadowH - 1) iy = u.ShadowH - 1;
        float stored = u.ShadowMap[(long)iy * u.ShadowW + ix];
        return d > Mul(stored, Add(1f, u.ShadowBias)) ? 0f : 1f;
    }
    static byte Unorm8(float x) { x = Min(Max(x, 0f), 1f); return (byte)Math.Floor(Add(Mul(x, 255f), 0.5f)); }

    static F3[] WorldCorners(Uniforms u)
    {
        F3[] c = new F3[4];
        for (int i = 0; i < 4; i++)
        {
            F4 v = Row(u.FrustumCorners, i); var v3 = new F3(v.x, v.y, v.z);
            F4 r0 = Row(u.CamInvView, 0), r1 = Row(u.CamInvView, 1), r2 = Row(u.CamInvView, 2);
            c[i] = new F3(Dot(new F3(r0.x, r0.y, r0.z), v3), Dot(new F3(r1.x, r1.y, r1.z), v3), Dot(new F3(r2.x, r2.y, r2.z), v3));
        }
        return c;
    }

    // shader:149-216 for every pixel; row 0 = bottom row.  src (RGBA8, may be null: Background) is _MainTex, sceneDepth (may be
    // null: FarPlane) the depth texture as linear eye depth (shader:154); steps (may be null) receives the evaluations of the
    // primary and of the shadow march per pixel.  Returns the hit count.
    public static long Render(Uniforms u, int w, int h, byte[] src, float[] sceneDepth, byte[] rgba, float[] depthOut, byte[] hitOut,
                              ushort[] steps = null)
    {
        F3[] c = WorldCorners(u);
        var cx = new Ctx(u);
        long hits = 0;
        Parallel.For(0, h, () => 0L, (y, _, local) =>
        {
            for (int x = 0; x < w; x++)
            {
                int idx = y * w + x;
                float uu = Div(Add(x, 0.5f), w), vv = Div(Add(y, 0.5f), h);
                F3 rd = Lerp(Lerp(c[3], c[2], uu), Lerp(c[0], c[1], uu), vv);
                March r = Raymarch(cx, u.CamPos, rd, sceneDepth != null ? sceneDepth[idx] : u.FarPlane);
                F3 col = src != null ? new F3(Div(src[4 * idx], 255f), Div(src[4 * idx + 1], 255f), Div(src[4 * idx + 2], 255f)) : u.Background;
                int ev = 0;
                if (r.hit)
                {
                    F3 hp = AddV(u.CamPos, Scale(rd, r.length));
                    float e = Max(0f, u.NormalSampleScale);
                    var raw = new F3(
                        Sub(Scene(cx, new F3(Add(hp.x, e), hp.y, hp.z)).w, Scene(cx, new F3(Sub(hp.x, e), hp.y, hp.z)).w),
                        Sub(Scene(cx, new F3(hp.x, Add(hp.y, e), hp.z)).w, Scene(cx, new F3(hp.x, Sub(hp.y, e), hp.z)).w),
                        Sub(Scene(cx, new F3(hp.x, hp.y, Add(hp.z, e))).w, Scene(cx, new F3(hp.x, hp.y, Sub(hp.z, e))).w));
                    F3 n = Normalize(raw);
                    F4 s = Scene(cx, hp);
                    float sh = u.ShadowMode != 0 ? ShadowLookup(u, hp, n, r.contact) : Shadow(cx, hp, n, r.contact, out ev);
                    float shadow = Add(Mul(Div(Sub(sh, 0f), Sub(1f, 0f)), Sub(1f, .25f)), .25f);                    // remap, cginc:80-82
                    float ndl = Add(Mul(Dot(n, u.Light), .5f), .5f);
                    float nss = u.NormalSampleScale;
                    float ao = Clamp(Len(new F3(Div(raw.x, nss), Div(raw.y, nss), Div(raw.z, nss))), 0f, 1f);
                    col = Scale(Scale(Scale(new F3(s.x, s.y, s.z), ndl), shadow), ao);
                    local++;
                }
                if (rgba != null) { rgba[4 * idx] = Unorm8(col.x); rgba[4 * idx + 1] = Unorm8(col.y); rgba[4 * idx + 2] = Unorm8(col.z); rgba[4 * idx + 3] = 255; }
                if (depthOut != null) depthOut[idx] = r.length;
                if (hitOut != null) hitOut[idx] = (byte)(r.hit ? 1 : 0);
                if (steps != null) { steps[2 * idx] = (ushort)r.evals; steps[2 * idx + 1] = (ushort)(r.hit ? ev : 0); }
            }
            return local;
        }, local => System.Threading.Interlocked.Add(ref hits, local));
        return hits;
    }

    // The depth passes (this library's definition: the reference's ShadowMap fragment is a stub): raymarch() from the shadow
    // camera through the pixel's ray (corners12 == null; ShadowMap.cs Render) or orthographically from the rectangle of
    // ShadowMapDirectional.cs:25-34 (corners TL, TR, BR, BL as 12 floats) along forward.  Writes r.length; returns the hit count.
    public static long RenderDepth(Uniforms u, int w, int h, float[] corners12, F3 forward, float[] depthOut, byte[] hitOut)
    {
        F3[] c = WorldCorners(u);
        var cx = new Ctx(u);
        F3[] dc = null;
        if (corners12 != null)
        {
            dc = new F3[4];
            for (int i = 0; i < 4; i++) dc[i] = new F3(corners12[3 * i], corners12[3 * i + 1], corners12[3 * i + 2]);
        }
        long hits = 0;
        Parallel.For(0, h, () => 0L, (y, _, local) =>
        {
            for (int x = 0; x < w; x++)
            {
                float uu = Div(Add(x, 0.5f), w), vv = Div(Add(y, 0.5f), h);
                F3 ro = u.CamPos, rd;
                if (dc == null) rd = Lerp(Lerp(c[3], c[2], uu), Lerp(c[0], c[1], uu), vv);
                else { ro = Lerp(Lerp(dc[3], dc[2], uu), Lerp(dc[0], dc[1], uu), vv); rd = forward; }
                March r = Raymarch(cx, ro, rd, u.FarPlane);
                int idx = y * w + x;
                if (depthOut != null) depthOut[idx] = r.length;
                if (hitOut != null) hitOut[idx] = (byte)(r.hit ? 1 : 0);
                if (r.hit) local++;
            }
            return local;
        }, local => System.Threading.Interlocked.Add(ref hits, local));
        return hits;
    }
}
