// render_kernels.hip - fused NeRF render pass for gfx950 (MI355X): ONE wavefront owns ONE ray.
//
// For each 32-sample tile of its ray the wave computes depths -> points -> positional
// encoding (VALU) -> [deformation MLP ->] canonical MLP (MFMA, registers: mlp_core.h) ->
// alpha compositing (wave scan), and after the last tile optionally the hierarchical
// resampling (inverse-CDF + rank merge in the wave's LDS slice).  Nothing per-sample
// touches HBM unless the caller asks for it (raw / weights / dx / z_out).
//
// Reference: render_rays nerf/run.py:316-422, d_nerf/run_dnerf.py:354-480; raw2outputs
// ray.py:155-198; sample_pdf ray.py:96-153; run_network nerf/run.py:73-87.
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "swnerf_common.h"
#include "mlp_core.h"
#include "host_util.h"
#include "mlp_kernels.h"

#define SW_LDS_SC 256                    // max coarse samples when resampling
#define SW_LDS_SORT 1024                 // max S + n_importance (padded to a power of two)
#define SW_LDS_WAVE_FLOATS (3 * SW_LDS_SC + SW_LDS_SORT)

struct PassDev {
    swnerf_pass_args a;
    const float* w0;        // weight stream this pass runs per tile
    const float* b0;        // its bias stream
    int nbias;              // floats in the bias stream (multiple of 32)
    int two_pass;           // 1: deformation net then canonical net
    int sort_n, sort_s;     // powers of two >= n_importance / >= n_samples (fallback sort of an unsorted list)
};

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

__device__ __forceinline__ float wave32_sum(float v) {   // sum over the 32 lanes of each half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
    return v;
}

__device__ __forceinline__ float z_linear(const swnerf_pass_args& a, float near, float far, int s) {
    const float t = sw_linspace(0.f, 1.f, a.n_samples, s);
    if (!a.lindisp) return near * (1.f - t) + far * t;                       // nerf/run.py:363
    return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);                   // nerf/run.py:365
}

// depth of sample s (0 <= s < S) of this ray
__device__ __forceinline__ float z_sample(const swnerf_pass_args& a, int64_t ray, float near, float far, int s) {
    const int S = a.n_samples;
    if (a.z_vals) return a.z_vals[ray * S + s];
    const float zs = z_linear(a, near, far, s);
    if (!a.t_rand) return zs;
    // stratified jitter, nerf/run.py:369-383
    const float upper = (s < S - 1) ? .5f * (z_linear(a, near, far, s + 1) + zs) : zs;
    const float lower = (s > 0) ? .5f * (zs + z_linear(a, near, far, s - 1)) : zs;
    return lower + (upper - lower) * a.t_rand[ray * S + s];
}

// ------------------------------------------------------------------------------------------
template <bool DNERF>
__global__ void __launch_bounds__(256, 1) render_pass_kernel(PassDev P) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const swnerf_pass_args& a = P.a;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
    bias_to_lds(lds_all, P.b0, P.nbias);         // the only block barrier; waves are independent after it
    if (ray >= a.n_rays) return;                 // wave-uniform
    const float* lds_bias = lds_all;
    float* lds_ring = lds_all + SW_LDS_BIAS_FLOATS + wv * SW_LDS_RING_FLOATS;
    float* lds_emb = lds_ring + SW_RING * SW_STEP_FLOATS;
    float* lds = lds_all + SW_LDS_FIXED_FLOATS + wv * SW_LDS_WAVE_FLOATS;
    float* zc = lds;                             // [S]   depths of this pass
    float* wc = lds + SW_LDS_SC;                 // [S]   compositing weights
    float* cdf = lds + 2 * SW_LDS_SC;            // [S-1]
    float* srt = lds + 3 * SW_LDS_SC;            // [sort_n]

    const int S = a.n_samples;
    const bool resample = a.n_importance > 0;
    const float* rb = a.ray_batch + ray * a.cols;
    const float ox = rb[0], oy = rb[1], oz = rb[2], dx = rb[3], dy = rb[4], dz = rb[5];
    const float near = rb[6], far = rb[7];
    const float ft = (a.cols == 12) ? rb[8] : 0.f;
    const float v0 = rb[a.cols - 3], v1 = rb[a.cols - 2], v2 = rb[a.cols - 1];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);                  // ray.py:173

    {   // once per ray: the view-direction encoding, parked in LDS (see tile_park)
        f32x16 demb;
        pe_dir(v0, v1, v2, h, demb);
        tile_park(lds_emb + 2 * 16 * 64, lane, demb);
    }
    WStream ws;
    ws_start(ws, P.w0, lds_bias, lds_ring, lane);

    const float* zrow = a.z_vals ? a.z_vals + ray * S : nullptr;
    const float* zslot = lds_emb + SW_EMB_LDS_FLOATS;
    const unsigned zslot_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)zslot);
    float pr = 0.f, pg = 0.f, pb = 0.f, pd = 0.f, pa = 0.f;
    double Tc = 1.0;                              // transmittance carried across tiles
    const int ntiles = (S + 31) >> 5;
#pragma nounroll
    for (int tile = 0; tile < ntiles; ++tile) {
        const int s = tile * 32 + j;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        float z, zn;
        if (zrow && tile > 0) {
            // depths given (fine pass): fetched by LDS-DMA while the previous tile's MLP ran - issued before that
            // tile's weight stream, so long landed - instead of a global load whose latency every tile would expose
            const float* zs = zslot + (tile & 1) * 64;
            z = zs[j];
            const float z1 = zs[j + 1];
            zn = (s + 1 < S) ? z1 : z;
        } else {
            z = z_sample(a, ray, near, far, sc);
            zn = (s + 1 < S) ? z_sample(a, ray, near, far, s + 1) : z;
        }
        if (zrow && tile + 1 < ntiles)
            lds_dma_dword(reinterpret_cast<const char*>(zrow), (unsigned)min(32 * (tile + 1) + lane, S - 1) * 4u,
                          zslot_addr + (unsigned)((tile + 1) & 1) * 256u);
        // pts = rays_o + rays_d * z  (two roundings, nerf/run.py:385)
        float px = ox + dx * z, py = oy + dy * z, pz = oz + dz * z;

        f32x16 emb[2], in[8], out[8];
        float head[3], rgb[3];
        pe_pos(px, py, pz, h, emb);
        if (DNERF) {
#pragma nounroll
            for (int pass = P.two_pass ? 0 : 1; pass < 2; ++pass) {
                trunk_pass<true>(emb, lds_emb, ft, pass == 0, h, in, out, head, ws);
                if (pass == 0) {
                    // dx = _time_out(h) (model.py:136,146-149)
                    const float ex = head[0], ey = head[1], ez = head[2];
                    if (a.dx && live && h == 0) {
                        float* o = a.dx + (ray * S + s) * 3;
                        o[0] = ex; o[1] = ey; o[2] = ez;
                    }
                    px = px + ex; py = py + ey; pz = pz + ez;
                    pe_pos(px, py, pz, h, emb);                              // re-embed (model.py:148-149)
                }
            }
            if (!P.two_pass && a.dx && live && h == 0) {
                float* o = a.dx + (ray * S + s) * 3;
                o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;                          // model.py:144-145
            }
        } else {
            trunk_pass<false>(emb, lds_emb, 0.f, false, h, in, out, head, ws);
        }
        {
            f32x16 demb;
            tile_fetch(lds_emb + 2 * 16 * 64, lane, demb);
            canon_tail(in, out, demb, rgb, ws.bias - SW_BIAS_TILE_FLOATS, ws);
        }
        ws_rewind(ws, P.w0, lds_bias, lane);

        // ---- raw2outputs on this tile (ray.py:155-198); both lane halves mirror each other
        const float c0 = rgb[0], c1 = rgb[1], c2 = rgb[2];
        float sg = head[0];
        if (a.raw && live && h == 0) {
            f32x4 r4 = {c0, c1, c2, sg};
            *reinterpret_cast<f32x4*>(a.raw + (ray * S + s) * 4) = r4;
        }
        if (a.noise) sg += a.noise[ray * S + sc];
        float dist = (s + 1 < S) ? (zn - z) : 1e10f;
        dist = dist * dnorm;
        float alpha = 1.f - expf(-fmaxf(sg, 0.f) * dist);
        if (!live) alpha = 0.f;
        double ps = (double)(1.f - alpha + 1e-10f);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const double up = __shfl_up(ps, o, 32);
            if (j >= o) ps *= up;
        }
        double ex = __shfl_up(ps, 1, 32);
        if (j == 0) ex = 1.0;
        const float T = (float)(Tc * ex);                                    // exclusive cumprod (ray.py:188)
        Tc *= __shfl(ps, 31, 32);
        const float w = alpha * T;
        if (live) {
            if (a.weights && h == 0) a.weights[ray * S + s] = w;
            if (a.z_out && h == 0) a.z_out[ray * S + s] = z;
            if (resample && h == 0) { zc[s] = z; wc[s] = w; }
        }
        pr += w * (1.f / (1.f + expf(-c0)));
        pg += w * (1.f / (1.f + expf(-c1)));
        pb += w * (1.f / (1.f + expf(-c2)));
        pd += w * z;
        pa += w;
    }

    pr = wave32_sum(pr); pg = wave32_sum(pg); pb = wave32_sum(pb);
    pd = wave32_sum(pd); pa = wave32_sum(pa);
    if (lane == 0) {
        if (a.rgb_map) {
            const float bg = a.white_bkgd ? (1.f - pa) : 0.f;                // ray.py:195-196
            a.rgb_map[ray * 3 + 0] = pr + bg;
            a.rgb_map[ray * 3 + 1] = pg + bg;
            a.rgb_map[ray * 3 + 2] = pb + bg;
        }
        if (a.depth_map) a.depth_map[ray] = pd;
        if (a.acc_map) a.acc_map[ray] = pa;
        if (a.disp_map) {
            const float q = pd / pa;                                         // NaN when acc == 0, kept (ray.py:192)
            a.disp_map[ray] = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
        }
    }
    if (!resample) return;

    // ---- sample_pdf (ray.py:96-153) on bins = mid-points, weights[1:-1]; then sort (nerf/run.py:396-400)
    wave_lds_sync();
    const int nb = S - 1, nw = S - 2, Ni = a.n_importance;
    double dpart = 0.0;                              // rounded once: see misc_kernels.hip sample_pdf_kernel
    for (int i = lane; i < nw; i += 64) dpart += (double)(wc[i + 1] + 1e-5f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dpart += __shfl_xor(dpart, o, 64);
    const float wsum = (float)dpart;
    double carry = 0.0;
    for (int base = 0; base < nw; base += 64) {      // cumsum accumulates in double like ATen's CPU kernel
        const int i = base + lane;
        double v = (i < nw) ? (double)((wc[i + 1] + 1e-5f) / wsum) : 0.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(v, o, 64);
            if (lane >= o) v += up;
        }
        if (i < nw) cdf[i + 1] = (float)(carry + v);
        carry += __shfl(v, 63, 64);
    }
    if (lane == 0) cdf[0] = 0.f;
    wave_lds_sync();
    double sm = 0.0;
    for (int m = lane; m < Ni; m += 64) {
        const float u = a.u ? a.u[ray * Ni + m] : sw_linspace(0.f, 1.f, Ni, m);
        int lo = 0, hi = nb;                         // searchsorted(cdf, u, right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(0, lo - 1), above = min(nb - 1, lo);
        const float cb = cdf[below], ca = cdf[above];
        const float bb = .5f * (zc[below + 1] + zc[below]), ba = .5f * (zc[above + 1] + zc[above]);
        float den = ca - cb;
        if (den < 1e-5f) den = 1.f;
        const float smp = bb + (u - cb) / den * (ba - bb);
        srt[m] = smp;
        sm += (double)smp;
    }
    wave_lds_sync();
    if (a.z_std) {                                   // torch.std(z_samples, unbiased=False), nerf/run.py:416
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        const double mean = sm / Ni;
        double var = 0.0;
        for (int m = lane; m < Ni; m += 64) { const double d = (double)srt[m] - mean; var += d * d; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
        if (lane == 0) a.z_std[ray] = (float)sqrt(var / Ni);
    }
    // ---- z_vals = sort(cat[z_vals, z_samples]) (nerf/run.py:400) as a MERGE of two sorted lists.
    // The coarse depths are sorted by construction (linspace, or jitter inside disjoint strata).  The samples are
    // sorted when u is (det: linspace; the inverse cdf is monotone) - up to a last-bit inversion where one bin ends
    // and the next begins, and not at all for random u - so that is CHECKED, and only an unsorted list is sorted
    // first (bitonic, on the Ni samples alone).  Then each element's slot = its own index + the number of elements
    // of the other list in front of it (ties: coarse depths first), found by binary search in the wave's LDS
    // slice: 13 dependent LDS reads per lane instead of the 36 barrier-separated stages of a 256-element bitonic
    // sort.  Any correct sort yields the same values as torch.sort.
    // (The coarse list is checked too: it arrives sorted except, in principle, for a last-bit inversion between two
    // jittered strata, or when a caller hands unsorted z_vals to a resampling pass.)
    wave_sort_if_unsorted(srt, Ni, P.sort_n, lane);
    wave_sort_if_unsorted(zc, S, P.sort_s, lane);
    float* zf = a.z_fine + ray * (S + Ni);
    for (int i = lane; i < S; i += 64) {             // coarse depth i goes behind the samples strictly below it
        const float v = zc[i];
        int lo = 0, hi = Ni;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (srt[mid] < v) lo = mid + 1; else hi = mid;
        }
        zf[i + lo] = v;
    }
    for (int m = lane; m < Ni; m += 64) {            // sample m goes behind the coarse depths <= it
        const float v = srt[m];
        int lo = 0, hi = S;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (zc[mid] <= v) lo = mid + 1; else hi = mid;
        }
        zf[m + lo] = v;
    }
}

// ------------------------------------------------------------------------------------------
// network_query_fn on bare points (nerf/load_model.py:56-74) with V view directions per point
// (nerf/extract_mesh.py:27-90): trunk + density once, view branch V times.
struct QueryDev {
    const float* pts; int64_t M; const float* dirs; int64_t V; int shared;
    const float* w0; const float* b0; int nbias; const float* wvl; float* out;
};

__global__ void __launch_bounds__(256, 1) query_points_kernel(QueryDev P) {
    extern __shared__ __attribute__((aligned(16))) float lds_bias[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds_ring = lds_bias + SW_LDS_BIAS_FLOATS + wv * SW_LDS_RING_FLOATS;
    float* lds_emb = lds_ring + SW_RING * SW_STEP_FLOATS;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    bias_to_lds(lds_bias, P.b0, P.nbias);
    if (tile * 32 >= P.M) return;
    const int64_t row = tile * 32 + j;
    const bool live = row < P.M;
    const int64_t rr = live ? row : P.M - 1;
    f32x16 emb[2], in[8], out[8];
    float head[3];
    pe_pos(P.pts[rr * 3], P.pts[rr * 3 + 1], P.pts[rr * 3 + 2], h, emb);
    WStream ws;
    ws_start(ws, P.w0, lds_bias, lds_ring, lane);
    trunk_pass<false>(emb, lds_emb, 0.f, false, h, in, out, head, ws);
    const float* hb_rgb = ws.bias - SW_BIAS_TILE_FLOATS;      // [b_alpha, b_r, b_g, b_b]
    seg_mfma<8, 8, SEG_BIAS>(out, in, ws);                   // feature = feature_linear(h)
    // the ring now holds the first VIEWS steps; the views-loop region starts with the same ones
    ws.base = reinterpret_cast<const char*>(P.wvl);
    float sr = 0.f, sg = 0.f, sb = 0.f;
    const int64_t nv = P.shared ? P.V : 1;
#pragma nounroll
    for (int64_t v = 0; v < nv; ++v) {
        const float* dp = P.dirs + (P.shared ? v : rr) * 3;
        f32x16 k9[9];
#pragma unroll
        for (int n = 0; n < 8; ++n) k9[n] = out[n];
        pe_dir(dp[0], dp[1], dp[2], h, k9[8]);
        ws.bias = lds_bias + SW_CANON_BIAS_TILE_VIEWS * SW_BIAS_TILE_FLOATS + h * 16;
        f32x16 hv[4];
        seg_mfma<4, 9, SEG_BIAS>(hv, k9, ws);
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[n][r] = relu1(hv[n][r]);
        float c3[3];
        head_valu<3, 4>(hv, ws, c3);
        sr += c3[0] + hb_rgb[1]; sg += c3[1] + hb_rgb[2]; sb += c3[2] + hb_rgb[3];
        ws.base = reinterpret_cast<const char*>(P.wvl);      // the region's tail is its own head
    }
    if (live && h == 0) {
        const float inv = 1.f / (float)nv;
        f32x4 r4 = {sr * inv, sg * inv, sb * inv, head[0]};
        *reinterpret_cast<f32x4*>(P.out + row * 4) = r4;
    }
}

extern "C" int swnerf_render_pass(const swnerf_pass_args* args, void* stream) {
    if (!args) return sw_fail(SWNERF_E_ARG, "render_pass: NULL args");
    const swnerf_pass_args& a = *args;
    if (!a.packed || (!a.ray_batch && a.n_rays != 0)) return sw_fail(SWNERF_E_ARG, "render_pass: NULL ray_batch/packed");
    if (a.n_rays < 0 || a.n_samples < 2) return sw_fail(SWNERF_E_ARG, "render_pass: n_rays %lld, n_samples %d", (long long)a.n_rays, a.n_samples);
    if (a.cols != 11 && a.cols != 12) return sw_fail(SWNERF_E_ARG, "render_pass: ray_batch must have 11 or 12 columns (use_viewdirs), got %d", a.cols);
    if (a.kind == SWNERF_NET_DNERF && a.cols != 12) return sw_fail(SWNERF_E_ARG, "render_pass: D-NeRF needs the frame_time column");
    if (a.L_pos < 0 || a.L_pos > 10 || a.L_dir < 0 || a.L_dir > 4 || a.L_time < 0 || a.L_time > 10)
        return sw_fail(SWNERF_E_UNSUPP, "render_pass: embedder bands (%d,%d,%d) exceed (10,4,10)", a.L_pos, a.L_dir, a.L_time);
    if (a.z_vals && a.t_rand) return sw_fail(SWNERF_E_ARG, "render_pass: t_rand only applies to coarse sampling");
    PassDev P;
    P.a = a;
    int rc = stream_ptrs(a.kind, a.packed, a.run_deform, &P.w0, &P.b0, &P.nbias, &P.two_pass);
    if (rc) return rc;
    P.sort_n = 0; P.sort_s = 0;
    size_t lds = SW_LDS_FIXED_FLOATS * sizeof(float);
    if (a.n_importance > 0) {
        if (!a.z_fine && a.n_rays != 0) return sw_fail(SWNERF_E_ARG, "render_pass: n_importance>0 needs z_fine");
        if (a.n_samples < 3 || a.n_samples > SW_LDS_SC || a.n_samples + a.n_importance > SW_LDS_SORT)
            return sw_fail(SWNERF_E_UNSUPP, "render_pass: resampling supports 3<=N_samples<=%d and N_samples+N_importance<=%d", SW_LDS_SC, SW_LDS_SORT);
        int p2 = 2;                              // fallback sort of an unsorted sample list: power of two >= n_importance
        while (p2 < a.n_importance) p2 <<= 1;
        P.sort_n = p2;
        p2 = 2;
        while (p2 < a.n_samples) p2 <<= 1;
        P.sort_s = p2;
        lds += 4 * SW_LDS_WAVE_FLOATS * sizeof(float);
    }
    if (a.n_rays == 0) return 0;
    const dim3 grid((unsigned)((a.n_rays + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (a.kind == SWNERF_NET_DNERF) hipLaunchKernelGGL(render_pass_kernel<true>, grid, block, lds, st, P);
    else {
        // a canonical-only net has no deformation: position_delta is zeros (NeRFOriginal.forward, model.py:273-296
        // returns torch.zeros_like(input_pts[:, :3])); the static kernel has no dx store, so fill it here
        if (a.dx) {
            rc = sw_check(hipMemsetAsync(a.dx, 0, (size_t)a.n_rays * a.n_samples * 3 * sizeof(float), st), "render_pass dx fill");
            if (rc) return rc;
        }
        hipLaunchKernelGGL(render_pass_kernel<false>, grid, block, lds, st, P);
    }
    return sw_check(hipGetLastError(), "render_pass launch");
}

// Coarse sampling alone (nerf/run.py:355-385; API parity for the op-by-op path - the fused pass does this in
// registers): z_vals [N,S] and, when asked, pts = o + d*z [N,S,3].  Same device functions as the fused pass.
__global__ void __launch_bounds__(256) sample_coarse_kernel(swnerf_pass_args a, float* z_out, float* pts) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int S = a.n_samples;
    if (idx >= a.n_rays * S) return;
    const int64_t ray = idx / S;
    const int s = (int)(idx - ray * S);
    const float* rb = a.ray_batch + ray * a.cols;
    const float z = z_sample(a, ray, rb[6], rb[7], s);
    z_out[idx] = z;
    if (pts) {
        pts[idx * 3 + 0] = rb[0] + rb[3] * z;
        pts[idx * 3 + 1] = rb[1] + rb[4] * z;
        pts[idx * 3 + 2] = rb[2] + rb[5] * z;
    }
}

extern "C" int swnerf_sample_coarse(const float* ray_batch, int64_t n_rays, int cols, int n_samples, int lindisp,
                                    const float* t_rand, float* z_vals, float* pts, void* stream) {
    if (n_rays == 0) return 0;                           // empty tensors have NULL data pointers
    if (!ray_batch || !z_vals || n_rays < 0 || n_samples < 1 || cols < 8)
        return sw_fail(SWNERF_E_ARG, "sample_coarse: NULL pointer, n_rays %lld, n_samples %d or cols %d < 8", (long long)n_rays, n_samples, cols);
    swnerf_pass_args a = {};
    a.ray_batch = ray_batch; a.n_rays = n_rays; a.cols = cols; a.n_samples = n_samples; a.lindisp = lindisp; a.t_rand = t_rand;
    const int64_t total = n_rays * n_samples;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, z_vals, pts);
    return sw_check(hipGetLastError(), "sample_coarse launch");
}

extern "C" int swnerf_query_points(const float* packed, const float* pts, int64_t M, const float* dirs, int64_t n_dirs,
                                   int shared_dirs, int L_pos, int L_dir, float* out, void* stream) {
    if (M == 0 && packed) return 0;
    if (!packed || !pts || !dirs || !out || M < 0) return sw_fail(SWNERF_E_ARG, "query_points: NULL pointer or negative M");
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "query_points: embedder bands (%d,%d) exceed (10,4)", L_pos, L_dir);
    if (shared_dirs ? n_dirs < 1 : n_dirs != M)
        return sw_fail(SWNERF_E_ARG, "query_points: need %s directions, got %lld for %lld points", shared_dirs ? ">= 1 shared" : "one per point", (long long)n_dirs, (long long)M);
    QueryDev P;
    P.pts = pts; P.M = M; P.dirs = dirs; P.V = n_dirs; P.shared = shared_dirs ? 1 : 0; P.out = out;
    P.w0 = packed; P.b0 = packed + SW_CANON_W_FLOATS; P.nbias = SW_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS;
    P.wvl = packed + SW_CANON_VL_OFFSET;
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    hipLaunchKernelGGL(query_points_kernel, grid, block, SW_LDS_FIXED_FLOATS * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "query_points launch");
}

extern "C" int swnerf_mlp_forward(int kind, const float* packed, const float* x, int64_t M, int L_pos, int L_dir,
                                  const float* t_emb, int L_time, int run_deform, float* out, float* dx_out, void* stream) {
    if (M == 0 && packed) return 0;
    if (!packed || !x || !out || M < 0) return sw_fail(SWNERF_E_ARG, "mlp_forward: NULL pointer or negative M");
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4 || L_time < 0 || L_time > 10)
        return sw_fail(SWNERF_E_UNSUPP, "mlp_forward: embedder bands (%d,%d,%d) exceed (10,4,10)", L_pos, L_dir, L_time);
    if (kind == SWNERF_NET_DNERF && run_deform && !t_emb) return sw_fail(SWNERF_E_ARG, "mlp_forward: D-NeRF deformation needs t_emb");
    MlpDev P;
    P.x = x; P.M = M; P.Lp = L_pos; P.Ld = L_dir; P.Lt = L_time;
    P.Cpos = 3 * (1 + 2 * L_pos); P.C = P.Cpos + 3 * (1 + 2 * L_dir);
    P.t_emb = t_emb; P.Ct = 1 + 2 * L_time;
    P.out = out; P.dx = dx_out; P.act = nullptr; P.bits = nullptr;
    int rc = stream_ptrs(kind, packed, run_deform, &P.w0, &P.b0, &P.nbias, &P.two_pass);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    const size_t lds = SW_LDS_FIXED_FLOATS * sizeof(float);
    if (kind == SWNERF_NET_DNERF) hipLaunchKernelGGL(mlp_forward_kernel<true>, grid, block, lds, st, P);
    else hipLaunchKernelGGL(mlp_forward_kernel<false>, grid, block, lds, st, P);
    return sw_check(hipGetLastError(), "mlp_forward launch");
}
