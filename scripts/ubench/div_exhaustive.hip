// Exhaustive check of x / s computed as  q0 = x*r; e = fma(-q0, s, x); q = fma(e, r, q0)  with
// r = RN(1/s), against IEEE division, for all binary32 x in [2^-60, 2^60] (both signs implied by
// symmetry) and a list of divisors s including adversarial significands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>

__global__ void check(uint32_t first, float s, float r, unsigned long long *bad, uint32_t *example) {
    uint32_t bits = first + blockIdx.x * blockDim.x + threadIdx.x;
    float x = __uint_as_float(bits);
    float ref = x / s;                       // correctly rounded by default in HIP
    float q0 = x * r;
    float e = __builtin_fmaf(-q0, s, x);
    float q = __builtin_fmaf(e, r, q0);
    if (__float_as_uint(ref) != __float_as_uint(q)) { atomicAdd(bad, 1ull); example[0] = bits; }
    if (__float_as_uint(ref) != __float_as_uint(q0)) atomicAdd(bad + 1, 1ull);      // sanity: the uncorrected product must differ often
}

int main() {
    unsigned long long *bad, h = 0, h0 = 0, tot0 = 0;
    uint32_t *ex, hex = 0;
    hipMalloc(&bad, 16); hipMalloc(&ex, 4);
    std::vector<float> ss = {0.5f, 1.5f, 0.7f, 1.3f, 0.8f, 1.1f, 1.9f, 0.6f, 1.2f, 1.05f, 0.9f, 3.0f, 7.0f, 0.3f, 0.1f, 1.6f,
                             1.0000001f, 1.9999999f, 0.99999994f, 1.9999998f, 1.5000001f, 1.4142135f, 0.70710677f, 2.7182817f,
                             3.1415927f, 1.3333334f, 0.6666667f, 1.1111112f, 0.123456789f, 123.456f, 1.75f, 1.25f, 1.125f, 1.0625f,
                             1.96875f, 1.984375f, 1.9921875f, 1.9990234f, 1.9999695f, 1.0078125f};
    // plus pseudo-random scales in [0.5, 1.5)
    uint32_t st = 12345;
    for (int i = 0; i < 60; i++) { st = st * 1664525u + 1013904223u; ss.push_back(0.5f + (st >> 8) / 16777216.0f); }
    uint32_t lo = (127u - 60u) << 23, hi = (127u + 60u) << 23;
    int worst = 0;
    for (float s : ss) {
        float r = 1.0f / s;
        hipMemset(bad, 0, 16);
        for (uint64_t b = lo; b < hi; b += (1u << 28)) {
            uint32_t n = (uint32_t)((hi - b) < (1u << 28) ? (hi - b) : (1u << 28));
            hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, (uint32_t)b, s, r, bad, ex);
        }
        hipDeviceSynchronize();
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&h0, bad + 1, 8, hipMemcpyDeviceToHost); tot0 += h0;
        hipMemcpy(&hex, ex, 4, hipMemcpyDeviceToHost);
        uint32_t sb; memcpy(&sb, &s, 4);
        if (h) { printf("s = %.9g (bits 0x%08x, significand 0x%06x): %llu mismatches, e.g. x bits 0x%08x\n", s, sb, sb & 0x7fffff, h, hex); worst++; }
    }
    printf("sanity: x*r alone differs from x/s in %llu cases\n", tot0);
    printf("%zu divisors checked over %llu x each: %d divisors with mismatches\n", ss.size(), (unsigned long long)(hi - lo), worst);
    return 0;
}
