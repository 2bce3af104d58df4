// resample.h - hierarchical resampling by ONE wavefront in its LDS slice: sample_pdf (ray.py:96-153), z_std
// (nerf/run.py:416) and z_vals = sort(cat[z_vals, z_samples]) (nerf/run.py:400) as a rank merge.
// Shared by the fused render pass (render_pass.h: the weights and depths are already in LDS) and the standalone op
// (misc_kernels.hip swnerf_sample_pdf: stages its operands into LDS first) - the SAME instruction sequence, so the two give
// the same bits (tests/test_gpu_parity.py::test_c2_full_size_properties holds them to torch.equal).
#pragma once
#include <hip/hip_runtime.h>
#include "swnerf_common.h"
#include "wave_dpp.h"

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Ascending in-place sort of buf[0..n) by one wave, ONLY if it is not already sorted (wave-uniform test); n_pow2 =
// power of two >= n, buf has room for n_pow2 floats (the pad is filled with +inf).  Bitonic network.
__device__ __forceinline__ void wave_sort_if_unsorted(float* buf, int n, int n_pow2, int lane) {
    bool sorted = true;
    for (int m = lane; m + 1 < n; m += 64) sorted = sorted && (buf[m] <= buf[m + 1]);
    if (__all(sorted)) return;
    for (int i = n + lane; i < n_pow2; i += 64) buf[i] = __builtin_inff();
    wave_lds_sync();
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int idx = lane; idx < (n_pow2 >> 1); idx += 64) {
                const int i = 2 * idx - (idx & (jj - 1));
                const int l = i + jj;
                const float x = buf[i], y = buf[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) { buf[i] = y; buf[l] = x; }
            }
            wave_lds_sync();
        }
    }
}

// bins of sample_pdf: the mid-points of the coarse depths (nerf/run.py:396) or a given array
struct MidBins { const float* z; __device__ __forceinline__ float operator()(int i) const { return .5f * (z[i + 1] + z[i]); } };
struct ArrayBins { const float* b; __device__ __forceinline__ float operator()(int i) const { return b[i]; } };

// sample_pdf for one ray.  w: the nb-1 weights (LDS), bins(i): bin edge i < nb, u_row: this ray's uniforms (global) or NULL
// for det (u = linspace(0, 1, Ni), ray.py:117-118), cdf: LDS scratch of nb floats, smp: LDS, Ni samples out.
// Returns this lane's partial sum of the samples in double (for z_std).  Ends with the samples visible to the wave.
//   pdf normaliser: sum(weights + 1e-5) accumulated in double and rounded once - the closest any order can get to ATen's
//   float sum (whose own blocking is machine dependent; DESIGN.md "conditioning"); cumsum in double like ATen's CPU kernel.
template <class Bins>
__device__ __forceinline__ double wave_sample_pdf(const float* w, int nb, Bins bins, const float* u_row, int Ni, float* cdf, float* smp, int lane) {
    const int nw = nb - 1;
    double dpart = 0.0;
    for (int i = lane; i < nw; i += 64) dpart += (double)(w[i] + 1e-5f);
    const float wsum = (float)wave_last_f64(wave_incl_sum_f64(dpart));
    double carry = 0.0;
    for (int base = 0; base < nw; base += 64) {
        const int i = base + lane;
        double v = (i < nw) ? (double)((w[i] + 1e-5f) / wsum) : 0.0;
        v = wave_incl_sum_f64(v);
        if (i < nw) cdf[i + 1] = (float)(carry + v);
        carry += wave_last_f64(v);
    }
    if (lane == 0) cdf[0] = 0.f;
    wave_lds_sync();
    double sm = 0.0;
    for (int m = lane; m < Ni; m += 64) {
        const float u = u_row ? u_row[m] : sw_linspace(0.f, 1.f, Ni, m);
        int lo = 0, hi = nb;                         // searchsorted(cdf, u, right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(0, lo - 1), above = min(nb - 1, lo);
        const float cb = cdf[below], ca = cdf[above];
        const float bb = bins(below), ba = bins(above);
        float den = ca - cb;
        if (den < 1e-5f) den = 1.f;                  // ray.py:148-149
        const float s = bb + (u - cb) / den * (ba - bb);
        smp[m] = s;
        sm += (double)s;
    }
    wave_lds_sync();
    return sm;
}

// torch.std(z_samples, unbiased=False) (nerf/run.py:416) from the per-lane partial sums of wave_sample_pdf; valid on every lane
__device__ __forceinline__ float wave_zstd(double sm, const float* smp, int Ni, int lane) {
    const double mean = wave_last_f64(wave_incl_sum_f64(sm)) / Ni;
    double var = 0.0;
    for (int m = lane; m < Ni; m += 64) { const double d = (double)smp[m] - mean; var += d * d; }
    return (float)sqrt(wave_last_f64(wave_incl_sum_f64(var)) / Ni);
}

// z_fine[0 .. S+Ni) = sort(cat[zc[0..S), smp[0..Ni)]) as a MERGE of two sorted lists (any correct sort yields torch.sort's
// values).  The coarse depths are sorted by construction (linspace, or jitter inside disjoint strata); the samples are when
// u is (det: linspace; the inverse cdf is monotone) - up to a last-bit inversion where one bin ends and the next begins, and
// not at all for random u - so both are CHECKED and only an unsorted list is sorted first (bitonic, on that list alone;
// sort_s / sort_n = powers of two >= S / Ni, the buffers hold that many floats).  Then each element's slot = its own index +
// the number of elements of the other list in front of it (ties: coarse depths first), by binary search in LDS: 13
// dependent LDS reads per lane instead of the 36 barrier-separated stages of a 256-element bitonic sort.
__device__ __forceinline__ void wave_rank_merge(float* zc, int S, int sort_s, float* smp, int Ni, int sort_n, float* zf, int lane) {
    wave_sort_if_unsorted(smp, Ni, sort_n, lane);
    wave_sort_if_unsorted(zc, S, sort_s, lane);
    for (int i = lane; i < S; i += 64) {             // coarse depth i goes behind the samples strictly below it
        const float v = zc[i];
        int lo = 0, hi = Ni;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (smp[mid] < v) lo = mid + 1; else hi = mid;
        }
        zf[i + lo] = v;
    }
    for (int m = lane; m < Ni; m += 64) {            // sample m goes behind the coarse depths <= it
        const float v = smp[m];
        int lo = 0, hi = S;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (zc[mid] <= v) lo = mid + 1; else hi = mid;
        }
        zf[m + lo] = v;
    }
}
