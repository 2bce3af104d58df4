// misc_kernels.hip - the HBM-bound satellites of the render path as standalone gfx950
// kernels (API parity with ray.py / embedder.py; the fused pass in render_kernels.hip does
// the same arithmetic in registers).  All are elementwise or one-wave-per-ray, coalesced.
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "swnerf_common.h"
#include "host_util.h"
#include "resample.h"

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
// A block packs 256 rays: the 11- / 12-float rows are assembled in LDS and leave as 16-byte stores of the block's
// contiguous 11 / 12 KiB of the batch (a row-per-thread store pattern writes 11 dwords at a 44-byte stride: every store
// instruction touches 22 lines for 256 useful bytes).  The last, partial block stores dword-wise.
__global__ void __launch_bounds__(256) pack_rays_kernel(const float* ro, const float* rd, int64_t n, float near, float far,
                                                        int has_time, float ft, int ndc, float sx, float sy, float* out) {
    __shared__ __attribute__((aligned(16))) float rows[256 * 12];
    const int t = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * 256, i = i0 + t;
    const int cols = has_time ? 12 : 11;
    if (i < n) {
        float ox = ro[i * 3], oy = ro[i * 3 + 1], oz = ro[i * 3 + 2], dx = rd[i * 3], dy = rd[i * 3 + 1], dz = rd[i * 3 + 2];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        const float v0 = dx / nrm, v1 = dy / nrm, v2 = dz / nrm;     // viewdirs BEFORE the NDC warp
        if (ndc) ndc_one(sx, sy, 1.f, ox, oy, oz, dx, dy, dz);       // caller hard-wires near=1. (nerf/run.py:149)
        float* o = rows + t * cols;
        o[0] = ox; o[1] = oy; o[2] = oz; o[3] = dx; o[4] = dy; o[5] = dz; o[6] = near; o[7] = far;
        int k = 8;
        if (has_time) o[k++] = ft;
        o[k] = v0; o[k + 1] = v1; o[k + 2] = v2;
    }
    __syncthreads();
    float* dst = out + i0 * cols;                                     // 256 * cols * 4 bytes per block: 16-byte aligned when `out` is
    const int live = (int)min((int64_t)256, n - i0) * cols;
    if (live == 256 * cols && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        for (int q = t; q < 64 * cols; q += 256) reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(rows)[q];
    } else {
        for (int q = t; q < live; q += 256) dst[q] = rows[q];
    }
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
// out[row] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], blocks d wide.  VALU-issue bound, not HBM
// bound, once the stores are coalesced: what counts is instructions per output float.  A workgroup builds the image of
// EMB_ROWS output rows in LDS - one job per (row, band, component) evaluates the sine AND the cosine of its argument with one
// reduction (sw_sincos_pair: the bits of sw_sin_or_cos), one job per (row, component) copies x - and the image, contiguous in
// the output and 16-byte aligned, leaves as float4 stores.  Job -> (row, slot) by a multiply-shift (q_magic = ceil(2^24 / Q),
// exact for the jobs of a block - checked on the host for every accepted (d, L); the 64-bit division per element of the round 1-3 kernel cost more than the sin/cos).
#define EMB_ROWS 64
__global__ void __launch_bounds__(256) embed_kernel(const float* x, int64_t M, int d, int L, int R, unsigned q_magic, unsigned d_magic, float* out) {
    extern __shared__ __attribute__((aligned(16))) float emb_img[];           // [rows][C]
    const int C = d * (1 + 2 * L), Q = d * (1 + L);                             // jobs per row: d copies + d * L (sin, cos) pairs
    const int64_t row0 = (int64_t)blockIdx.x * R;                             // R <= EMB_ROWS rows per workgroup (fewer for very wide rows: LDS)
    const int rows = (int)min((int64_t)R, M - row0);
    const float* xb = x + row0 * d;
    for (unsigned j = threadIdx.x; j < (unsigned)(rows * Q); j += 256) {
        const unsigned row = (j * q_magic) >> 24, q = j - row * (unsigned)Q;
        float* o = emb_img + row * C;
        if (q < (unsigned)d) {
            o[q] = xb[row * d + q];
        } else {
            const unsigned p = q - d, k = (p * d_magic) >> 24, c = p - k * (unsigned)d;
            float sv, cv;
            sw_sincos_pair(xb[row * d + c] * (float)(1 << k), &sv, &cv);    // x * 2^k is exact (embedder.py:29,36)
            o[d + 2 * k * d + c] = sv;
            o[d + 2 * k * d + d + c] = cv;
        }
    }
    __syncthreads();
    float* dst = out + row0 * C;
    const int total = rows * C;
    if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        for (int i = threadIdx.x; i < total / 4; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(emb_img)[i];
    } else {
        for (int i = threadIdx.x; i < total; i += 256) dst[i] = emb_img[i];
    }
}

extern "C" int swnerf_embed(const float* x, int64_t M, int d, int L, float* out, void* stream) {
    if (M == 0 && d > 0 && L >= 0) return 0;
    if (!x || !out || M < 0 || d <= 0 || d > 16 || L < 0 || L > 24) return sw_fail(SWNERF_E_ARG, "embed: bad arguments (M=%lld d=%d L=%d; d <= 16, L <= 24)", (long long)M, d, L);
    const int C = d * (1 + 2 * L), Q = d * (1 + L);
    const unsigned q_magic = ((1u << 24) + Q - 1) / Q, d_magic = ((1u << 24) + d - 1) / d;   // floor(n / Q) = (n * q_magic) >> 24 for n * Q < 2^24
    int R = EMB_ROWS;
    while (R > 1 && (size_t)R * C * sizeof(float) > 64 * 1024) R >>= 1;
    hipLaunchKernelGGL(embed_kernel, dim3(nblocks(M, R)), dim3(256), (size_t)R * C * sizeof(float), (hipStream_t)stream,
                       x, M, d, L, R, q_magic, d_magic, out);
    return sw_check(hipGetLastError(), "embed launch");
}

// ---- raw2outputs (ray.py:155-198): one wave per ray, 64 samples per sweep ------------------------
// Loads: raw as one float4 per lane (1 KiB per wave instruction), z one dword per lane; the next sample's depth comes from the
// lane above (DPP), only lane 63 loads it.  The exclusive cumprod runs in double like ATen's CPU cumprod, on the DPP path
// (wave_dpp.h: this kernel is VALU-issue bound - 3 sigmoids and an exp per sample - so a scan step must not cost eight
// instructions and two LDS round trips); the transmittance carried from sweep to sweep is wave-uniform.
__global__ void __launch_bounds__(256) raw2outputs_kernel(const float* raw, const float* zv, const float* rd, const float* noise,
                                                          int64_t N, int S, int white, float* rgb_map, float* disp, float* acc,
                                                          float* weights, float* depth) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= N) return;
    const float dx = rd[ray * 3], dy = rd[ray * 3 + 1], dz = rd[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float* zr = zv + ray * S;
    const float4* rr = reinterpret_cast<const float4*>(raw) + ray * S;
    const float* nr = noise ? noise + ray * S : nullptr;
    float* wr = weights ? weights + ray * S : nullptr;
    float pr = 0.f, pg = 0.f, pb = 0.f, pd = 0.f, pa = 0.f;
    double Tc = 1.0;                                      // wave-uniform
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        const float4 r4 = rr[sc];
        const float z = zr[sc];
        const float z_edge = (lane == 63 && s + 1 < S) ? zr[s + 1] : 0.f;
        const float zn = wave_from_above_f32(z, z_edge);
        float dist = (s + 1 < S) ? (zn - z) : 1e10f;
        dist = dist * dnorm;
        float sg = r4.w;
        if (nr) sg += nr[sc];
        float alpha = 1.f - expf(-fmaxf(sg, 0.f) * dist);
        if (!live) alpha = 0.f;
        const double ps = wave_incl_prod_f64((double)(1.f - alpha + 1e-10f));
        const double ex = wave_from_below_f64(ps, 1.0);
        const float w = alpha * (float)(Tc * ex);
        Tc *= wave_last_f64(ps);
        if (live && wr) wr[s] = w;
        pr += w * (1.f / (1.f + expf(-r4.x)));
        pg += w * (1.f / (1.f + expf(-r4.y)));
        pb += w * (1.f / (1.f + expf(-r4.z)));
        pd += w * z;
        pa += w;
    }
    pr = wave_sum_to_last_f32(pr); pg = wave_sum_to_last_f32(pg); pb = wave_sum_to_last_f32(pb);
    pd = wave_sum_to_last_f32(pd); pa = wave_sum_to_last_f32(pa);
    if (lane == 63) {
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
// One wave per ray, four rays per workgroup, each wave in its own LDS slice (no block barrier anywhere): the operands are staged
// into LDS with coalesced loads, then the wave-level routines of resample.h run - the code of the fused render pass, bit for bit.
// Slice layout (floats): w[nb-1 -> nbp] | bins[nbp] | cdf[nbp] | z[sort_s] | samples[sort_n]   (nbp = nb rounded up to 4)
#define SP_MAX_BINS 1024
#define SP_MAX_SORT 2048
__global__ void __launch_bounds__(256) sample_pdf_kernel(const float* bins, const float* wts, int64_t N, int nb, int ns,
                                                         const float* u_in, float* samples, const float* zv, int S,
                                                         float* z_sorted, float* z_std, int sort_s, int sort_n, int slice) {
    extern __shared__ __attribute__((aligned(16))) float sp_lds[];
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= N) return;                                  // wave-uniform
    const int nbp = (nb + 3) & ~3;
    float* w = sp_lds + (threadIdx.x >> 6) * slice;
    float* b = w + nbp;
    float* cdf = b + nbp;
    float* z = cdf + nbp;
    float* smp = z + sort_s;
    for (int i = lane; i < nb - 1; i += 64) w[i] = wts[ray * (nb - 1) + i];
    for (int i = lane; i < nb; i += 64) b[i] = bins[ray * nb + i];
    if (z_sorted) for (int i = lane; i < S; i += 64) z[i] = zv[ray * S + i];
    wave_lds_sync();
    const double sm = wave_sample_pdf(w, nb, ArrayBins{b}, u_in ? u_in + ray * ns : nullptr, ns, cdf, smp, lane);
    for (int m = lane; m < ns; m += 64) samples[ray * ns + m] = smp[m];      // in draw order (ray.py:150-151), before any sort
    if (z_std) {
        const float sd = wave_zstd(sm, smp, ns, lane);
        if (lane == 0) z_std[ray] = sd;
    }
    if (z_sorted) wave_rank_merge(z, S, sort_s, smp, ns, sort_n, z_sorted + ray * (S + ns), lane);
}

extern "C" int swnerf_sample_pdf(const float* bins, const float* weights, int64_t N, int nb, int n_samples, const float* u,
                                 float* samples, const float* z_vals, int S, float* z_sorted, float* z_std, void* stream) {
    if (N == 0 && nb >= 2 && n_samples >= 1) return 0;
    if (!bins || !weights || !samples || N < 0 || nb < 2 || n_samples < 1)
        return sw_fail(SWNERF_E_ARG, "sample_pdf: bad arguments (N=%lld nb=%d n_samples=%d)", (long long)N, nb, n_samples);
    if (nb > SP_MAX_BINS) return sw_fail(SWNERF_E_UNSUPP, "sample_pdf: at most %d bins", SP_MAX_BINS);
    if (n_samples > SP_MAX_SORT) return sw_fail(SWNERF_E_UNSUPP, "sample_pdf: at most %d samples per ray", SP_MAX_SORT);
    int sort_s = 0, sort_n = 2;
    while (sort_n < n_samples) sort_n <<= 1;               // the sample buffer doubles as the fallback sort's power-of-two pad
    if (z_sorted) {
        if (!z_vals || S < 1 || S + n_samples > SP_MAX_SORT) return sw_fail(SWNERF_E_UNSUPP, "sample_pdf: sort needs z_vals and S+n_samples <= %d", SP_MAX_SORT);
        sort_s = 2;
        while (sort_s < S) sort_s <<= 1;
    }
    if (N == 0) return 0;
    const int nbp = (nb + 3) & ~3;
    const int slice = 3 * nbp + sort_s + sort_n;
    const size_t lds = (size_t)4 * slice * sizeof(float);  // <= 4 x (3 x 1024 + 2048 + 2048) x 4 B = 112 KiB
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(nblocks(N, 4)), dim3(256), lds, (hipStream_t)stream, bins, weights, N, nb, n_samples, u,
                       samples, z_vals, S, z_sorted, z_std, sort_s, sort_n, slice);
    return sw_check(hipGetLastError(), "sample_pdf launch");
}
