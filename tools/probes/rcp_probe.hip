// rcp_probe.hip -- accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 (max relative error over 2^24 operands spread over a wide range of exponents), and of the refinements built on
// them: one / two Newton steps, one cubic step y (1 + e + e^2).  Build: hipcc --offload-arch=gfx950 -O3 -o rcp_probe rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void probe(double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // operand: mantissa from a hash of i, exponent sweeping [-300, 300]
    unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const double m = 1. + (double)(h >> 11) * (1. / 9007199254740992.);
    const double d = ldexp(m, (int)(h % 601) - 300);
    const double y0 = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y0, 1.);
    const double y1 = fma(e, y0, y0);
    double e1 = fma(-d, y1, 1.);
    const double y2 = fma(e1, y1, y1);
    const double t = fma(e, e, e);
    const double yc = fma(y0, t, y0);
    out[5 * (size_t)i + 0] = d; out[5 * (size_t)i + 1] = y0; out[5 * (size_t)i + 2] = y1; out[5 * (size_t)i + 3] = y2; out[5 * (size_t)i + 4] = yc;
}

int main() {
    const int n = 1 << 22;
    double* dev; hipMalloc(&dev, sizeof(double) * 5 * (size_t)n);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dev, n);
    std::vector<double> h(5 * (size_t)n); hipMemcpy(h.data(), dev, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    long double worst[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double d = h[5 * (size_t)i], exact = 1.0L / d;
        for (int k = 0; k < 4; ++k) { const long double err = fabsl((long double)h[5 * (size_t)i + 1 + k] - exact) / exact; if (err > worst[k]) worst[k] = err; }
    }
    printf("v_rcp_f64: max relative error %.3Le = 2^%.1Lf\n", worst[0], log2l(worst[0]));
    printf("  + one Newton step:   %.3Le = 2^%.1Lf\n", worst[1], log2l(worst[1]));
    printf("  + two Newton steps:  %.3Le = 2^%.1Lf\n", worst[2], log2l(worst[2]));
    printf("  + one cubic step y (1 + e + e^2): %.3Le = 2^%.1Lf\n", worst[3], log2l(worst[3]));
    return 0;
}
