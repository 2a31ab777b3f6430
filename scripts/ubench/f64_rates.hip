// Microbenchmark: sustained issue rate of the fp64 VALU instructions the canonical transcendental series are made of
// (cycles per wave64 instruction at 1, 2, 4, 8 waves per SIMD, 8 independent chains per lane), next to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 2048
#define KERNEL(name, decl, body, sink)                                         \
__global__ __launch_bounds__(256) void name(double *out, double a, double b) { \
    decl;                                                                      \
    for (int i = 0; i < ITER; i++) { body; }                                   \
    out[blockIdx.x * 256 + threadIdx.x] = sink;                                \
}
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DECLD(i) double x##i = a + i + threadIdx.x;
#define SUMD (x0+x1+x2+x3+x4+x5+x6+x7)
#define DECLF(i) float f##i = (float)a + i + threadIdx.x;
#define SUMF (double)(f0+f1+f2+f3+f4+f5+f6+f7)

#define ADD(i)   asm volatile("v_add_f64 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define MUL(i)   asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define FMA(i)   asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
#define MAX(i)   asm volatile("v_max_f64 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define RCP(i)   asm volatile("v_rcp_f64 %0, %0" : "+v"(x##i));
#define RSQ(i)   asm volatile("v_rsq_f64 %0, %0" : "+v"(x##i));
#define SQRT(i)  asm volatile("v_sqrt_f64 %0, %0" : "+v"(x##i));
#define FLOOR(i) asm volatile("v_floor_f64 %0, %0" : "+v"(x##i));
#define LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x##i));
#define CMP(i)   asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(x##i), "v"(a) : "vcc");
#define DSCALE(i) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(x##i) : "v"(a) : "vcc");
#define DFMAS(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b) : "vcc");
#define DFIX(i)  asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
#define CVTDF(i) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(x##i) : "v"(f##i));
#define CVTFD(i) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(f##i) : "v"(x##i));
#define FMAF(i)  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f##i) : "v"(fa), "v"(fb));
#define ADDS(i)  asm volatile("v_add_f64 %0, %0, %1" : "+v"(x##i) : "s"(a));
#define DIV(i)   x##i = a / x##i;
#define SQRTC(i) x##i = __builtin_sqrt(x##i);

KERNEL(k_add, R8(DECLD), R8(ADD), SUMD)
KERNEL(k_mul, R8(DECLD), R8(MUL), SUMD)
KERNEL(k_fma, R8(DECLD), R8(FMA), SUMD)
KERNEL(k_max, R8(DECLD), R8(MAX), SUMD)
KERNEL(k_rcp, R8(DECLD), R8(RCP), SUMD)
KERNEL(k_rsq, R8(DECLD), R8(RSQ), SUMD)
KERNEL(k_sqrt, R8(DECLD), R8(SQRT), SUMD)
KERNEL(k_floor, R8(DECLD), R8(FLOOR), SUMD)
KERNEL(k_ldexp, R8(DECLD), R8(LDEXP), SUMD)
KERNEL(k_cmp, R8(DECLD), R8(CMP), SUMD)
KERNEL(k_dscale, R8(DECLD), R8(DSCALE), SUMD)
KERNEL(k_dfmas, R8(DECLD), R8(DFMAS), SUMD)
KERNEL(k_dfix, R8(DECLD), R8(DFIX), SUMD)
KERNEL(k_cvtdf, R8(DECLD) R8(DECLF), R8(CVTDF), SUMD)
KERNEL(k_cvtfd, R8(DECLD) R8(DECLF), R8(CVTFD), SUMF)
KERNEL(k_fmaf, float fa = (float)a; float fb = (float)b; R8(DECLF), R8(FMAF), SUMF)
KERNEL(k_adds, R8(DECLD), R8(ADDS), SUMD)
KERNEL(k_div, R8(DECLD), R8(DIV), SUMD)          // the compiler's IEEE fp64 divide, one per chain and iteration
KERNEL(k_sqrtc, R8(DECLD), R8(SQRTC), SUMD)      // the compiler's correctly rounded fp64 sqrt

typedef void (*kfn)(double *, double, double);
struct Entry { const char *name; kfn fn; };

int main() {
    double *out;
    const int blocks_per_cu[] = {1, 2, 4, 8};
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    Entry es[] = {{"v_fma_f32 (reference)", k_fmaf}, {"v_add_f64", k_add}, {"v_add_f64 sgpr src", k_adds}, {"v_mul_f64", k_mul}, {"v_fma_f64", k_fma}, {"v_max_f64", k_max},
                  {"v_cmp_lt_f64", k_cmp}, {"v_floor_f64", k_floor}, {"v_ldexp_f64", k_ldexp}, {"v_cvt_f64_f32", k_cvtdf}, {"v_cvt_f32_f64", k_cvtfd},
                  {"v_rcp_f64", k_rcp}, {"v_rsq_f64", k_rsq}, {"v_sqrt_f64", k_sqrt},
                  {"v_div_scale_f64", k_dscale}, {"v_div_fmas_f64", k_dfmas}, {"v_div_fixup_f64", k_dfix},
                  {"a / x (IEEE divide, whole)", k_div}, {"sqrt(x) (correctly rounded)", k_sqrtc}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-30s", "instruction");
    for (int b : blocks_per_cu) printf("  %2d wv/SIMD: cyc/inst", b);
    printf("\n");
    for (auto &e : es) {
        printf("%-30s", e.name);
        for (int b : blocks_per_cu) {
            int grid = 256 * b;
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 1.0001, 0.5);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 1.0001, 0.5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double t = ms * 1e-3 / 5;
            double insts_per_simd = (double)ITER * 8 * b;
            printf("        %12.2f", t * 2.4e9 / insts_per_simd);
        }
        printf("\n");
    }
    return 0;
}
