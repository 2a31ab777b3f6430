// Checks that the double-precision operations the canonical transcendental functions are built from
// (sqrt, divide; add/mul are exact by construction) give the same bits on the GPU as on the host:
// 2^24 pseudo-random operands per range, several ranges.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__host__ __device__ inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// a double with a random 52-bit significand and an exponent in [-span, span]
__host__ __device__ inline double operand(uint64_t i, uint64_t salt, int span) {
    const uint64_t r = mix(i * 2 + salt);
    const int e = (int)(mix(r) % (uint64_t)(2 * span + 1)) - span;
    const uint64_t bits = ((uint64_t)(1023 + e) << 52) | (r & 0xFFFFFFFFFFFFFull);
    double d;
#ifdef __HIP_DEVICE_COMPILE__
    d = __longlong_as_double((long long)bits);
#else
    std::memcpy(&d, &bits, 8);
#endif
    return d;
}

__global__ void run(double *sq, double *dv, uint64_t n, int span) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = operand(i, 1, span), b = operand(i, 2, span);
    sq[i] = sqrt(a);
    dv[i] = a / b;
}

int main() {
    const uint64_t n = 1ull << 24;
    double *d_sq, *d_dv;
    if (hipMalloc(&d_sq, n * 8) != hipSuccess || hipMalloc(&d_dv, n * 8) != hipSuccess) return 2;
    std::vector<double> sq(n), dv(n);
    unsigned long long bad_sq = 0, bad_dv = 0;
    for (int span : {0, 8, 60, 500}) {
        hipLaunchKernelGGL(run, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d_sq, d_dv, n, span);
        if (hipMemcpy(sq.data(), d_sq, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        if (hipMemcpy(dv.data(), d_dv, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        for (uint64_t i = 0; i < n; i++) {
            const double a = operand(i, 1, span), b = operand(i, 2, span);
            const double s = std::sqrt(a), q = a / b;
            if (std::memcmp(&s, &sq[i], 8) != 0) bad_sq++;
            if (std::memcmp(&q, &dv[i], 8) != 0) bad_dv++;
        }
        std::printf("exponent span +-%d: %llu operands, sqrt mismatches %llu, divide mismatches %llu\n", span,
                    (unsigned long long)n, bad_sq, bad_dv);
    }
    return (bad_sq || bad_dv) ? 1 : 0;
}
