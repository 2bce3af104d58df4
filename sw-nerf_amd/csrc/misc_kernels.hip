// misc_kernels.hip - the HBM-bound satellites of the render path as standalone gfx950
// kernels (API parity with ray.py / embedder.py; the fused pass in render_kernels.hip does
// the same arithmetic in registers).  All are elementwise or one-wave-per-ray, coalesced.
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "swnerf_common.h"
#include "host_util.h"

static thread_local char g_err[SW_ERRBUF_LEN] = "";
char* sw_errbuf() { return g_err; }
extern "C" const char* swnerf_last_error(void) { return g_err; }
extern "C" int swnerf_version(void) { return SWNERF_VERSION; }
extern "C" size_t swnerf_packed_floats(int kind) {
    return kind == SWNERF_NET_CANON ? (size_t)SW_CANON_FLOATS
         : (kind == SWNERF_NET_DNERF ? (size_t)SW_DNERF_FLOATS : (kind == SWNERF_NET_NOVIEW ? (size_t)SW_NOVIEW_FLOATS : 0));
}

static inline unsigned nblocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// ---- get_rays (ray.py:10-38) ---------------------------------------------------------------
struct Cam { float fx, fy, cx, cy; float r[9]; float t[3]; };

__global__ void __launch_bounds__(256) get_rays_kernel(Cam c, int W, int64_t ray0, int64_t n, float* ro, float* rd) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t p = ray0 + i;
    const float px = (float)(p % W), py = (float)(p / W);        // integer pixel centres, no +0.5
    const float a = (px - c.cx) / c.fx, b = -(py - c.cy) / c.fy, m = -1.f;
    // sum(dirs[..., None, :] * c2w[:3,:3], -1): products rounded, then added left to right
    rd[i * 3 + 0] = a * c.r[0] + b * c.r[1] + m * c.r[2];
    rd[i * 3 + 1] = a * c.r[3] + b * c.r[4] + m * c.r[5];
    rd[i * 3 + 2] = a * c.r[6] + b * c.r[7] + m * c.r[8];
    if (ro) { ro[i * 3 + 0] = c.t[0]; ro[i * 3 + 1] = c.t[1]; ro[i * 3 + 2] = c.t[2]; }
}

extern "C" int swnerf_get_rays(int H, int W, double fx, double fy, double cx, double cy, int focal_branch,
                               const float* c2w, int64_t ray0, int64_t n, float* rays_o, float* rays_d, void* stream) {
    if (n == 0 && H > 0 && W > 0) return 0;
    if (!c2w || !rays_d || H <= 0 || W <= 0 || n < 0 || ray0 < 0 || ray0 + n > (int64_t)H * W)
        return sw_fail(SWNERF_E_ARG, "get_rays: bad arguments (H=%d W=%d ray0=%lld n=%lld)", H, W, (long long)ray0, (long long)n);
    Cam c;
    if (focal_branch) { c.fx = (float)fx; c.fy = (float)fx; c.cx = (float)(W * 0.5); c.cy = (float)(H * 0.5); }
    else { c.fx = (float)fx; c.fy = (float)fy; c.cx = (float)cx; c.cy = (float)cy; }
    for (int i = 0; i < 3; ++i) { for (int k = 0; k < 3; ++k) c.r[i * 3 + k] = c2w[i * 4 + k]; c.t[i] = c2w[i * 4 + 3]; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(get_rays_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, c, W, ray0, n, rays_o, rays_d);
    return sw_check(hipGetLastError(), "get_rays launch");
}

// ---- ndc_rays (ray.py:75-92) -----------------------------------------------------------------
__device__ __forceinline__ void ndc_one(float sx, float sy, float near, float& ox, float& oy, float& oz,
                                        float& dx, float& dy, float& dz) {
    const float t = -(near + oz) / dz;
    ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz;
    const float o0 = sx * ox / oz, o1 = sy * oy / oz, o2 = 1.f + 2.f * near / oz;
    const float d0 = sx * (dx / dz - ox / oz), d1 = sy * (dy / dz - oy / oz), d2 = -2.f * near / oz;
    ox = o0; oy = o1; oz = o2; dx = d0; dy = d1; dz = d2;
}

__global__ void __launch_bounds__(256) ndc_kernel(float sx, float sy, float near, const float* ro, const float* rd,
                                                  int64_t n, float* oo, float* od) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float ox = ro[i * 3], oy = ro[i * 3 + 1], oz = ro[i * 3 + 2], dx = rd[i * 3], dy = rd[i * 3 + 1], dz = rd[i * 3 + 2];
    ndc_one(sx, sy, near, ox, oy, oz, dx, dy, dz);
    oo[i * 3] = ox; oo[i * 3 + 1] = oy; oo[i * 3 + 2] = oz;
    od[i * 3] = dx; od[i * 3 + 1] = dy; od[i * 3 + 2] = dz;
}

// the python scalars -1./(W/(2.*focal)) are evaluated in double and then cast (ray.py:81-86)
static inline float ndc_scale(int WH, double focal) { return (float)(-1. / (WH / (2. * focal))); }

extern "C" int swnerf_ndc_rays(int H, int W, double focal, double near, const float* rays_o, const float* rays_d,
                               int64_t n, float* o_out, float* d_out, void* stream) {
    if (n == 0) return 0;
    if (!rays_o || !rays_d || !o_out || !d_out || n < 0) return sw_fail(SWNERF_E_ARG, "ndc_rays: NULL pointer / negative n");
    hipLaunchKernelGGL(ndc_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       ndc_scale(W, focal), ndc_scale(H, focal), (float)near, rays_o, rays_d, n, o_out, d_out);
    return sw_check(hipGetLastError(), "ndc_rays launch");
}

// ---- ray batch packing (nerf/run.py:137-158, d_nerf/run_dnerf.py:137-160) ----------------------
__global__ void __launch_bounds__(256) pack_rays_kernel(const float* ro, const float* rd, int64_t n, float near, float far,
                                                        int has_time, float ft, int ndc, float sx, float sy, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float ox = ro[i * 3], oy = ro[i * 3 + 1], oz = ro[i * 3 + 2], dx = rd[i * 3], dy = rd[i * 3 + 1], dz = rd[i * 3 + 2];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float v0 = dx / nrm, v1 = dy / nrm, v2 = dz / nrm;     // viewdirs BEFORE the NDC warp
    if (ndc) ndc_one(sx, sy, 1.f, ox, oy, oz, dx, dy, dz);       // caller hard-wires near=1. (nerf/run.py:149)
    const int cols = has_time ? 12 : 11;
    float* o = out + i * cols;
    o[0] = ox; o[1] = oy; o[2] = oz; o[3] = dx; o[4] = dy; o[5] = dz; o[6] = near; o[7] = far;
    int k = 8;
    if (has_time) o[k++] = ft;
    o[k] = v0; o[k + 1] = v1; o[k + 2] = v2;
}

extern "C" int swnerf_pack_ray_batch(const float* rays_o, const float* rays_d, int64_t n, double near, double far,
                                     int has_time, double frame_time, int ndc, int H, int W, double ndc_focal,
                                     float* ray_batch, void* stream) {
    if (n == 0) return 0;
    if (!rays_o || !rays_d || !ray_batch || n < 0) return sw_fail(SWNERF_E_ARG, "pack_ray_batch: NULL pointer / negative n");
    const float sx = ndc ? ndc_scale(W, ndc_focal) : 0.f, sy = ndc ? ndc_scale(H, ndc_focal) : 0.f;
    hipLaunchKernelGGL(pack_rays_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, n,
                       (float)near, (float)far, has_time, (float)frame_time, ndc, sx, sy, ray_batch);
    return sw_check(hipGetLastError(), "pack_ray_batch launch");
}

// ---- Embedder.embed (embedder.py:33-42) --------------------------------------------------------
// one thread per OUTPUT element so the [M, d(1+2L)] rows are written fully coalesced
__global__ void __launch_bounds__(256) embed_kernel(const float* x, int64_t total, int d, int C, float* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int64_t row = e / C;
    const int col = (int)(e - row * C);
    const int blk = col / d, c = col - blk * d;
    const float v = x[row * d + c];
    float r = v;
    if (blk > 0) {
        const int k = (blk - 1) >> 1;
        r = sw_sin_or_cos(v * (float)(1 << k), (blk - 1) & 1);
    }
    out[e] = r;
}

extern "C" int swnerf_embed(const float* x, int64_t M, int d, int L, float* out, void* stream) {
    if (M == 0 && d > 0 && L >= 0) return 0;
    if (!x || !out || M < 0 || d <= 0 || L < 0 || L > 24) return sw_fail(SWNERF_E_ARG, "embed: bad arguments (M=%lld d=%d L=%d)", (long long)M, d, L);
    const int C = d * (1 + 2 * L);
    const int64_t total = M * C;
    hipLaunchKernelGGL(embed_kernel, dim3(nblocks(total, 256)), dim3(256), 0, (hipStream_t)stream, x, total, d, C, out);
    return sw_check(hipGetLastError(), "embed launch");
}

// ---- raw2outputs (ray.py:155-198): one wave per ray, 64 samples per sweep ------------------------
__global__ void __launch_bounds__(256) raw2outputs_kernel(const float* raw, const float* zv, const float* rd, const float* noise,
                                                          int64_t N, int S, int white, float* rgb_map, float* disp, float* acc,
                                                          float* weights, float* depth) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= N) return;
    const float dx = rd[ray * 3], dy = rd[ray * 3 + 1], dz = rd[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    float pr = 0.f, pg = 0.f, pb = 0.f, pd = 0.f, pa = 0.f;
    double Tc = 1.0;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        const float4 r4 = *reinterpret_cast<const float4*>(raw + (ray * S + sc) * 4);
        const float z = zv[ray * S + sc];
        float dist = (s + 1 < S) ? (zv[ray * S + s + 1] - z) : 1e10f;
        dist = dist * dnorm;
        float sg = r4.w;
        if (noise) sg += noise[ray * S + sc];
        float alpha = 1.f - expf(-fmaxf(sg, 0.f) * dist);
        if (!live) alpha = 0.f;
        double ps = (double)(1.f - alpha + 1e-10f);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(ps, o, 64);
            if (lane >= o) ps *= up;
        }
        double ex = __shfl_up(ps, 1, 64);
        if (lane == 0) ex = 1.0;
        const float w = alpha * (float)(Tc * ex);
        Tc *= __shfl(ps, 63, 64);
        if (live && weights) weights[ray * S + s] = w;
        pr += w * (1.f / (1.f + expf(-r4.x)));
        pg += w * (1.f / (1.f + expf(-r4.y)));
        pb += w * (1.f / (1.f + expf(-r4.z)));
        pd += w * z;
        pa += w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        pr += __shfl_xor(pr, o, 64); pg += __shfl_xor(pg, o, 64); pb += __shfl_xor(pb, o, 64);
        pd += __shfl_xor(pd, o, 64); pa += __shfl_xor(pa, o, 64);
    }
    if (lane == 0) {
        if (rgb_map) {
            const float bg = white ? (1.f - pa) : 0.f;
            rgb_map[ray * 3] = pr + bg; rgb_map[ray * 3 + 1] = pg + bg; rgb_map[ray * 3 + 2] = pb + bg;
        }
        if (depth) depth[ray] = pd;
        if (acc) acc[ray] = pa;
        if (disp) { const float q = pd / pa; disp[ray] = 1.f / ((q != q) ? q : fmaxf(1e-10f, q)); }
    }
}

extern "C" int swnerf_raw2outputs(const float* raw, const float* z_vals, const float* rays_d, const float* noise, int64_t N, int S,
                                  int white_bkgd, float* rgb_map, float* disp_map, float* acc_map, float* weights,
                                  float* depth_map, void* stream) {
    if (S == 1) return sw_fail(SWNERF_E_UNSUPP, "raw2outputs: S=1 is degenerate in the reference (ray.py:170-171 builds an EMPTY dists tensor); need S>=2");
    if (N == 0 && S >= 2) return 0;
    if (!raw || !z_vals || !rays_d || N < 0 || S < 2) return sw_fail(SWNERF_E_ARG, "raw2outputs: bad arguments (N=%lld S=%d)", (long long)N, S);
    hipLaunchKernelGGL(raw2outputs_kernel, dim3(nblocks(N, 4)), dim3(256), 0, (hipStream_t)stream, raw, z_vals, rays_d, noise, N, S,
                       white_bkgd, rgb_map, disp_map, acc_map, weights, depth_map);
    return sw_check(hipGetLastError(), "raw2outputs launch");
}

// ---- sample_pdf (ray.py:96-153) [+ sort(cat[z_vals, samples]), nerf/run.py:400] -------------------
#define SP_MAX_BINS 1024
#define SP_MAX_SORT 2048
__global__ void __launch_bounds__(64) sample_pdf_kernel(const float* bins, const float* wts, int64_t N, int nb, int ns,
                                                        const float* u_in, float* samples, const float* zv, int S,
                                                        float* z_sorted, float* z_std, int sort_n) {
    __shared__ float cdf[SP_MAX_BINS];
    __shared__ float srt[SP_MAX_SORT];
    const int lane = threadIdx.x;
    const int64_t ray = blockIdx.x;
    const float* b = bins + ray * nb;
    const float* w = wts + ray * (nb - 1);
    const int nw = nb - 1;
    // sum(weights + 1e-5): accumulated in double and rounded once - the closest any order can get
    // to ATen's float sum (whose own blocking is machine dependent); see DESIGN.md "conditioning"
    double dpart = 0.0;
    for (int i = lane; i < nw; i += 64) dpart += (double)(w[i] + 1e-5f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dpart += __shfl_xor(dpart, o, 64);
    const float part = (float)dpart;
    double carry = 0.0;
    for (int base = 0; base < nw; base += 64) {
        const int i = base + lane;
        double v = (i < nw) ? (double)((w[i] + 1e-5f) / part) : 0.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(v, o, 64);
            if (lane >= o) v += up;
        }
        if (i < nw) cdf[i + 1] = (float)(carry + v);
        carry += __shfl(v, 63, 64);
    }
    if (lane == 0) cdf[0] = 0.f;
    __syncthreads();
    double sm = 0.0;
    for (int m = lane; m < ns; m += 64) {
        const float u = u_in ? u_in[ray * ns + m] : sw_linspace(0.f, 1.f, ns, m);
        int lo = 0, hi = nb;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(0, lo - 1), above = min(nb - 1, lo);
        const float cb = cdf[below], ca = cdf[above];
        float den = ca - cb;
        if (den < 1e-5f) den = 1.f;
        const float smp = b[below] + (u - cb) / den * (b[above] - b[below]);
        samples[ray * ns + m] = smp;
        if (z_sorted) srt[S + m] = smp;
        sm += (double)smp;
    }
    if (z_std) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        const double mean = sm / ns;
        double var = 0.0;
        for (int m = lane; m < ns; m += 64) { const double d = (double)samples[ray * ns + m] - mean; var += d * d; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
        if (lane == 0) z_std[ray] = (float)sqrt(var / ns);
    }
    if (!z_sorted) return;
    for (int i = lane; i < S; i += 64) srt[i] = zv[ray * S + i];
    for (int i = S + ns + lane; i < sort_n; i += 64) srt[i] = __builtin_inff();
    __syncthreads();
    for (int k = 2; k <= sort_n; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int idx = lane; idx < (sort_n >> 1); idx += 64) {
                const int i = 2 * idx - (idx & (jj - 1)), l = i + jj;
                const float x = srt[i], y = srt[l];
                if ((x > y) == ((i & k) == 0)) { srt[i] = y; srt[l] = x; }
            }
            __syncthreads();
        }
    for (int i = lane; i < S + ns; i += 64) z_sorted[ray * (S + ns) + i] = srt[i];
}

extern "C" int swnerf_sample_pdf(const float* bins, const float* weights, int64_t N, int nb, int n_samples, const float* u,
                                 float* samples, const float* z_vals, int S, float* z_sorted, float* z_std, void* stream) {
    if (N == 0 && nb >= 2 && n_samples >= 1) return 0;
    if (!bins || !weights || !samples || N < 0 || nb < 2 || n_samples < 1)
        return sw_fail(SWNERF_E_ARG, "sample_pdf: bad arguments (N=%lld nb=%d n_samples=%d)", (long long)N, nb, n_samples);
    if (nb > SP_MAX_BINS) return sw_fail(SWNERF_E_UNSUPP, "sample_pdf: at most %d bins", SP_MAX_BINS);
    int sort_n = 0;
    if (z_sorted) {
        if (!z_vals || S < 1 || S + n_samples > SP_MAX_SORT) return sw_fail(SWNERF_E_UNSUPP, "sample_pdf: sort needs z_vals and S+n_samples <= %d", SP_MAX_SORT);
        sort_n = 2;
        while (sort_n < S + n_samples) sort_n <<= 1;
    }
    if (N == 0) return 0;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)N), dim3(64), 0, (hipStream_t)stream, bins, weights, N, nb, n_samples, u,
                       samples, z_vals, S, z_sorted, z_std, sort_n);
    return sw_check(hipGetLastError(), "sample_pdf launch");
}
