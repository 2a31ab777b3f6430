// Microbenchmark: sustained issue rate of the VALU instructions the march is made of.
// Each kernel runs ITER iterations of 8 independent instruction chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 4096
typedef float float2_ __attribute__((ext_vector_type(2)));

#define KERNEL(name, decl, body, sink)                                         \
__global__ __launch_bounds__(256) void name(float *out, float a, float b) {   \
    decl;                                                                      \
    for (int i = 0; i < ITER; i++) { body; }                                   \
    out[blockIdx.x * 256 + threadIdx.x] = sink;                                \
}

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DECLF(i) float x##i = a + i + threadIdx.x;
#define SUMF (x0+x1+x2+x3+x4+x5+x6+x7)

#define FMA(i)  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
#define ADD(i)  asm volatile("v_add_f32 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define MUL(i)  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define MAX(i)  asm volatile("v_max_f32 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x##i));
#define RSQ(i)  asm volatile("v_rsq_f32 %0, %0" : "+v"(x##i));
#define RCP(i)  asm volatile("v_rcp_f32 %0, %0" : "+v"(x##i));
#define CND(i)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x##i) : "v"(a));
#define CMP(i)  asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x##i), "v"(a) : "vcc");
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
#define SUBABS(i) asm volatile("v_sub_f32_e64 %0, |%0|, %1" : "+v"(x##i) : "v"(a));
#define FMAS(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##i) : "s"(a), "v"(b));

KERNEL(k_fma, R8(DECLF), R8(FMA), SUMF)
KERNEL(k_add, R8(DECLF), R8(ADD), SUMF)
KERNEL(k_mul, R8(DECLF), R8(MUL), SUMF)
KERNEL(k_max, R8(DECLF), R8(MAX), SUMF)
KERNEL(k_sqrt, R8(DECLF), R8(SQRT), SUMF)
KERNEL(k_rsq, R8(DECLF), R8(RSQ), SUMF)
KERNEL(k_rcp, R8(DECLF), R8(RCP), SUMF)
KERNEL(k_cnd, R8(DECLF), R8(CND), SUMF)
KERNEL(k_cmp, R8(DECLF), R8(CMP), SUMF)
KERNEL(k_addu, R8(DECLF), R8(ADDU), SUMF)
KERNEL(k_max3, R8(DECLF), R8(MAX3), SUMF)
KERNEL(k_subabs, R8(DECLF), R8(SUBABS), SUMF)
KERNEL(k_fma_sgpr, R8(DECLF), R8(FMAS), SUMF)

#define DECLP(i) float2_ p##i; p##i.x = a + i + threadIdx.x; p##i.y = b + i;
#define SUMP (p0.x+p1.x+p2.x+p3.x+p4.x+p5.x+p6.x+p7.x+p0.y+p1.y+p2.y+p3.y+p4.y+p5.y+p6.y+p7.y)
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##i) : "v"(pa), "v"(pb));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(pa));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##i) : "v"(pa));
#define DPA float2_ pa; pa.x = a; pa.y = a; float2_ pb; pb.x = b; pb.y = b;
KERNEL(k_pkfma, DPA R8(DECLP), R8(PKFMA), SUMP)
KERNEL(k_pkmul, DPA R8(DECLP), R8(PKMUL), SUMP)
KERNEL(k_pkadd, DPA R8(DECLP), R8(PKADD), SUMP)


#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
#define FMAD(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##i) : "v"(y##i), "v"(z##i));
#define DECLYZ(i) float y##i = a * (i + 2), z##i = b * (i + 3);
#define MIN(i)  asm volatile("v_min_f32 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
#define CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "s"(m));
#define CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x##i) : "v"(a), "v"(b) : "vcc");
#define CMPCND64(i) asm volatile("v_cmp_lt_f32_e64 %3, %0, %1\n v_cndmask_b32_e64 %0, %0, %2, %3" : "+v"(x##i) : "v"(a), "v"(b), "s"(m));
#define CMP64(i) asm volatile("v_cmp_lt_f32_e64 %2, %0, %1" :: "v"(x##i), "v"(a), "s"(m));
#define CNDVCCW(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x##i) : "v"(a));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x##i) : "v"(a), "v"(b));
#define ANDB(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define MULS(i)  asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x##i) : "s"(a));
#define ADDC(i)  asm volatile("v_add_f32 %0, 0.5, %0" : "+v"(x##i));
#define MAXC(i)  asm volatile("v_max_f32 %0, 0, %0" : "+v"(x##i));
KERNEL(k_fmac, R8(DECLF), R8(FMAC), SUMF)
KERNEL(k_fmad, R8(DECLF) R8(DECLYZ), R8(FMAD), SUMF)
KERNEL(k_min, R8(DECLF), R8(MIN), SUMF)
KERNEL(k_med3, R8(DECLF), R8(MED3), SUMF)
KERNEL(k_cnd64, R8(DECLF) unsigned long long m = __ballot(threadIdx.x & 1), R8(CND64), SUMF)
KERNEL(k_cmpcnd, R8(DECLF), R8(CMPCND), SUMF)
KERNEL(k_cmpcnd64, R8(DECLF) unsigned long long m = __ballot(threadIdx.x & 1), R8(CMPCND64), SUMF)
KERNEL(k_cmp64, R8(DECLF) unsigned long long m = __ballot(threadIdx.x & 1), R8(CMP64), SUMF)
KERNEL(k_bfi, R8(DECLF), R8(BFI), SUMF)
KERNEL(k_and, R8(DECLF), R8(ANDB), SUMF)
KERNEL(k_muls, R8(DECLF), R8(MULS), SUMF)
KERNEL(k_addc, R8(DECLF), R8(ADDC), SUMF)
KERNEL(k_maxc, R8(DECLF), R8(MAXC), SUMF)


#define NOIEEE asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");
KERNEL(k_max_noieee, NOIEEE R8(DECLF), R8(MAX), SUMF)
KERNEL(k_min_noieee, NOIEEE R8(DECLF), R8(MIN), SUMF)
KERNEL(k_max3_noieee, NOIEEE R8(DECLF), R8(MAX3), SUMF)
KERNEL(k_muls_noieee, NOIEEE R8(DECLF), R8(MULS), SUMF)
KERNEL(k_cmp_noieee, NOIEEE R8(DECLF), R8(CMP), SUMF)
KERNEL(k_add_noieee, NOIEEE R8(DECLF), R8(ADD), SUMF)
#define MAXI(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x##i) : "v"(a));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x##i) : "v"(y##i), "v"(z##i));
#define LSHR(i) asm volatile("v_lshrrev_b32 %0, 31, %0" : "+v"(x##i));
#define SUBCL(i) asm volatile("v_sub_u32_e64 %0, %0, 1 clamp" : "+v"(x##i));
#define MULD(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x##i) : "v"(y##i));
#define MAXD(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x##i) : "v"(y##i));
KERNEL(k_maxi, R8(DECLF), R8(MAXI), SUMF)
KERNEL(k_add3, R8(DECLF) R8(DECLYZ), R8(ADD3), SUMF)
KERNEL(k_lshr, R8(DECLF), R8(LSHR), SUMF)
KERNEL(k_subcl, R8(DECLF), R8(SUBCL), SUMF)
KERNEL(k_muld, R8(DECLF) R8(DECLYZ), R8(MULD), SUMF)
KERNEL(k_maxd, R8(DECLF) R8(DECLYZ), R8(MAXD), SUMF)

// dependent chain: one accumulator
#define FMA1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
KERNEL(k_fma_dep, float x0 = a + threadIdx.x, FMA1 FMA1 FMA1 FMA1 FMA1 FMA1 FMA1 FMA1, x0)

typedef void (*kfn)(float *, float, float);
struct Entry { const char *name; kfn fn; int flops_per_inst; };

int main() {
    float *out;
    const int blocks_per_cu[] = {1, 2, 4, 8};
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 2);
    Entry es[] = {{"v_fma_f32", k_fma, 2}, {"v_fma_f32 (sgpr src)", k_fma_sgpr, 2}, {"v_add_f32", k_add, 1}, {"v_mul_f32", k_mul, 1}, {"v_max_f32", k_max, 1},
                  {"v_max3_f32", k_max3, 2}, {"v_sub_f32 |x|", k_subabs, 1}, {"v_cndmask_b32", k_cnd, 0}, {"v_cmp_lt_f32", k_cmp, 0}, {"v_add_u32", k_addu, 0},
                  {"v_sqrt_f32", k_sqrt, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_rcp_f32", k_rcp, 1},
                  {"v_pk_fma_f32", k_pkfma, 4}, {"v_pk_mul_f32", k_pkmul, 2}, {"v_pk_add_f32", k_pkadd, 2},
                  {"v_fma_f32 dependent chain", k_fma_dep, 2},
 {"v_fmac_f32", k_fmac, 2}, {"v_fma_f32 distinct regs", k_fmad, 2}, {"v_min_f32", k_min, 1}, {"v_med3_f32", k_med3, 1}, {"v_cndmask e64 sgpr mask", k_cnd64, 0}, {"v_cmp+v_cndmask vcc (pair)", k_cmpcnd, 0}, {"v_cmp+v_cndmask e64 (pair)", k_cmpcnd64, 0}, {"v_cmp e64 sgpr dst", k_cmp64, 0}, {"v_bfi_b32", k_bfi, 0}, {"v_and_b32", k_and, 0}, {"v_mul_f32 sgpr src", k_muls, 1}, {"v_add_f32 inline const", k_addc, 1}, {"v_max_f32 inline 0", k_maxc, 1},
 {"v_max_f32 IEEE=0", k_max_noieee, 1}, {"v_min_f32 IEEE=0", k_min_noieee, 1}, {"v_max3_f32 IEEE=0", k_max3_noieee, 1}, {"v_mul_f32 sgpr IEEE=0", k_muls_noieee, 1}, {"v_cmp_lt_f32 IEEE=0", k_cmp_noieee, 1}, {"v_add_f32 IEEE=0", k_add_noieee, 1}, {"v_max_i32", k_maxi, 0}, {"v_add3_u32 distinct", k_add3, 0}, {"v_lshrrev_b32", k_lshr, 0}, {"v_sub_u32 clamp", k_subcl, 0}, {"v_mul_f32 distinct", k_muld, 1}, {"v_max_f32 distinct", k_maxd, 1}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s", "instruction");
    for (int b : blocks_per_cu) printf("  %2d wv/SIMD: cyc/inst  Tinst/s", b);
    printf("\n");
    for (auto &e : es) {
        printf("%-28s", e.name);
        for (int b : blocks_per_cu) {
            int grid = 256 * b;
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double t = ms * 1e-3 / 5;
            double insts_per_simd = (double)ITER * 8 * b;        // wave-instructions per SIMD (b waves per SIMD)
            double cyc = t * 2.4e9 / insts_per_simd;              // at nominal 2.4 GHz
            double tinst = insts_per_simd * 1024 / t / 1e12;
            printf("        %6.2f  %7.3f", cyc, tinst);
        }
        printf("\n");
    }
    return 0;
}
