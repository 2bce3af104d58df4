// dpp_scan_test.hip - the wave-level DPP helpers of csrc/wave_dpp.h against serial host arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -I sw-nerf_amd/csrc tools/microbench/dpp_scan_test.hip -o /tmp/dpp_scan_test && /tmp/dpp_scan_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "wave_dpp.h"

__global__ void k(const double* in, double* prod, double* sum, double* below, float* above, double* last, float* fsum) {
    const int l = threadIdx.x;
    const double v = in[l];
    prod[l] = wave_incl_prod_f64(v);
    sum[l] = wave_incl_sum_f64(v);
    below[l] = wave_from_below_f64(v, -7.0);
    above[l] = wave_from_above_f32((float)v, -9.f);
    last[l] = wave_last_f64(v);
    fsum[l] = wave_sum_to_last_f32((float)v);
    if (l == 0) { /* a second use with a live subset must not disturb the others */ }
}

int main() {
    std::vector<double> h(64);
    for (int i = 0; i < 64; ++i) h[i] = 0.5 + 0.013 * i + (i % 7) * 0.11;
    double *d_in, *d_p, *d_s, *d_b, *d_l; float *d_a, *d_f;
    hipMalloc(&d_in, 512); hipMalloc(&d_p, 512); hipMalloc(&d_s, 512); hipMalloc(&d_b, 512); hipMalloc(&d_l, 512); hipMalloc(&d_a, 256); hipMalloc(&d_f, 256);
    hipMemcpy(d_in, h.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_p, d_s, d_b, d_a, d_l, d_f);
    std::vector<double> p(64), s(64), b(64), la(64); std::vector<float> a(64), f(64);
    hipMemcpy(p.data(), d_p, 512, hipMemcpyDeviceToHost); hipMemcpy(s.data(), d_s, 512, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d_b, 512, hipMemcpyDeviceToHost); hipMemcpy(la.data(), d_l, 512, hipMemcpyDeviceToHost);
    hipMemcpy(a.data(), d_a, 256, hipMemcpyDeviceToHost); hipMemcpy(f.data(), d_f, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    double cp = 1.0, cs = 0.0; float fs = 0.f;
    for (int i = 0; i < 64; ++i) {
        cp *= h[i]; cs += h[i]; fs += (float)h[i];
        if (fabs(p[i] - cp) > 1e-12 * fabs(cp)) { printf("prod[%d] %g vs %g\n", i, p[i], cp); ++bad; }
        if (fabs(s[i] - cs) > 1e-12 * fabs(cs)) { printf("sum[%d] %g vs %g\n", i, s[i], cs); ++bad; }
        const double wb = i ? h[i - 1] : -7.0;
        if (b[i] != wb) { printf("below[%d] %g vs %g\n", i, b[i], wb); ++bad; }
        const float wa = i < 63 ? (float)h[i + 1] : -9.f;
        if (a[i] != wa) { printf("above[%d] %g vs %g\n", i, a[i], wa); ++bad; }
        if (la[i] != h[63]) { printf("last[%d] %g vs %g\n", i, la[i], h[63]); ++bad; }
    }
    if (fabsf(f[63] - fs) > 1e-4f) { printf("fsum %g vs %g\n", f[63], fs); ++bad; }
    printf(bad ? "FAILED: %d mismatches\n" : "dpp helpers ok (%d)\n", bad);
    return bad ? 1 : 0;
}
