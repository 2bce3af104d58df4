// backward_kernels.hip - gradients of the render path.
//
//  * raw2outputs backward: d(raw) from d(rgb_map, disp_map, acc_map, depth_map, weights)   (ray.py:155-198)
//  * TN GEMM  dW[o][i] += sum_m A[m][o] * B[m][i]  (+ column sums for the bias gradient): the weight
//    gradients of every Linear layer of the 8x256 MLP, v_mfma_f32_32x32x2_f32, K = rows
// (the register-resident dX chain lives in render_kernels.hip next to the forward it mirrors)
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "swnerf_common.h"
#include "host_util.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// raw2outputs backward, one wave per ray.  With c = sigmoid(rgb), e = exp(-relu(sigma)*dist),
// a = 1-e, p = 1-a+1e-10, T_i = prod_{j<i} p_j, w = a*T:
//   G_i   = dL/dw_i = g_rgb.c_i + gA + gD*z_i + g_w_i
//   dL/da_i = G_i*T_i - (sum_{k>i} G_k*w_k)/p_i ,   da/dsigma = dist*e*[sigma>0]
//   dL/drgb_i = w_i * g_rgb * c_i*(1-c_i)
// gA folds d(acc_map), the white-background term (rgb_map += 1-acc) and disp = 1/max(1e-10, D/A);
// gD folds d(depth_map) and disp.  Prefix products and suffix sums run in double like the forward.
#define R2B_SMAX 1024
__global__ void __launch_bounds__(256) raw2outputs_bwd_kernel(const float* raw, const float* zv, const float* rd, const float* noise,
                                                              int64_t N, int S, int white, const float* g_rgb, const float* g_disp,
                                                              const float* g_acc, const float* g_depth, const float* g_w, float* d_raw) {
    __shared__ float sT[4][R2B_SMAX], sW[4][R2B_SMAX], sG[4][R2B_SMAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
    if (ray >= N) return;
    float* T_ = sT[wv]; float* W_ = sW[wv]; float* G_ = sG[wv];
    const float dx = rd[ray * 3], dy = rd[ray * 3 + 1], dz = rd[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float gr = g_rgb ? g_rgb[ray * 3] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    // pass 1: forward recompute of T, w; accumulate acc and depth for the disparity term
    double Tc = 1.0;
    float pa = 0.f, pd = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        const float z = zv[ray * S + sc];
        float dist = (s + 1 < S) ? (zv[ray * S + s + 1] - z) : 1e10f;
        dist *= dnorm;
        float sg = raw[(ray * S + sc) * 4 + 3];
        if (noise) sg += noise[ray * S + sc];
        float alpha = 1.f - expf(-fmaxf(sg, 0.f) * dist);
        if (!live) alpha = 0.f;
        double ps = (double)(1.f - alpha + 1e-10f);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double up = __shfl_up(ps, o, 64); if (lane >= o) ps *= up; }
        double ex = __shfl_up(ps, 1, 64);
        if (lane == 0) ex = 1.0;
        const float T = (float)(Tc * ex);
        Tc *= __shfl(ps, 63, 64);
        const float w = alpha * T;
        if (live) { T_[s] = T; W_[s] = w; }
        pa += w; pd += w * z;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pa += __shfl_xor(pa, o, 64); pd += __shfl_xor(pd, o, 64); }
    float gA = g_acc ? g_acc[ray] : 0.f, gD = g_depth ? g_depth[ray] : 0.f;
    if (white) gA -= (gr + gg + gb);
    if (g_disp) {
        const float q = pd / pa;                       // disp = 1/max(1e-10, q); no gradient on the clamped / NaN branch
        if (q > 1e-10f) { const float gq = -g_disp[ray] / (q * q); gD += gq / pa; gA -= gq * pd / (pa * pa); }
    }
    // pass 2: G_i, then the suffix sums of G*w from the last chunk to the first
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double carry = 0.0;
    const int nch = (S + 63) / 64;
    for (int ch = nch - 1; ch >= 0; --ch) {
        const int s = ch * 64 + lane;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(raw + (ray * S + sc) * 4);
        const float z = zv[ray * S + sc];
        const float c0 = 1.f / (1.f + expf(-r4[0])), c1 = 1.f / (1.f + expf(-r4[1])), c2 = 1.f / (1.f + expf(-r4[2]));
        const float w = live ? W_[sc] : 0.f, T = live ? T_[sc] : 0.f;
        float G = gr * c0 + gg * c1 + gb * c2 + gA + gD * z;
        if (g_w) G += g_w[ray * S + sc];
        // inclusive suffix sum of G*w over lanes >= this one, in double
        double v = live ? (double)G * (double)w : 0.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double dn = __shfl_down(v, o, 64); if (lane + o < 64) v += dn; }
        double after = __shfl_down(v, 1, 64);          // sum over lanes > this one in the chunk
        if (lane == 63) after = 0.0;
        const double R = carry + after;
        carry += __shfl(v, 0, 64);
        float dist = (s + 1 < S) ? (zv[ray * S + s + 1] - z) : 1e10f;
        dist *= dnorm;
        float sg = r4[3];
        if (noise) sg += noise[ray * S + sc];
        const float e = expf(-fmaxf(sg, 0.f) * dist);
        const float p = 1.f - (1.f - e) + 1e-10f;
        const float dLda = G * T - (float)(R / (double)p);
        const float dsig = (sg > 0.f) ? dLda * dist * e : 0.f;
        if (live) {
            f32x4 o4 = {w * gr * c0 * (1.f - c0), w * gg * c1 * (1.f - c1), w * gb * c2 * (1.f - c2), dsig};
            *reinterpret_cast<f32x4*>(d_raw + (ray * S + s) * 4) = o4;
        }
    }
}

extern "C" int swnerf_raw2outputs_backward(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                                           int64_t N, int S, int white_bkgd, const float* g_rgb, const float* g_disp,
                                           const float* g_acc, const float* g_depth, const float* g_weights, float* d_raw,
                                           void* stream) {
    if (S < 2 || S > R2B_SMAX) return sw_fail(SWNERF_E_UNSUPP, "raw2outputs_backward: 2 <= S <= %d (got %d)", R2B_SMAX, S);
    if (N == 0) return 0;
    if (!raw || !z_vals || !rays_d || !d_raw || N < 0) return sw_fail(SWNERF_E_ARG, "raw2outputs_backward: NULL pointer / negative N");
    hipLaunchKernelGGL(raw2outputs_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw, z_vals, rays_d,
                       noise, N, S, white_bkgd, g_rgb, g_disp, g_acc, g_depth, g_weights, d_raw);
    return sw_check(hipGetLastError(), "raw2outputs_backward launch");
}

// ---------------------------------------------------------------------------------------------
// C[o][i] += sum_m A[m][o] * B[m][i]   (A [M,lda] -> No columns, B [M,ldb] -> Ni columns, C [No, ldc]),
// bias[o] += sum_m A[m][o].   The weight gradient of one Linear layer: A = d(pre-activation), B = the
// layer's input, K = the (ray,sample) rows.  v_mfma_f32_32x32x2_f32 with k = two rows per step:
// lane (i, h') supplies A[m0+2s+h'][o0+i] and B[m0+2s+h'][i0+i] - both row-major operands are read
// as two contiguous 128-B segments per wave load.  One wave owns a 32 x 256 strip of C (8 accumulators);
// the workgroup's waves share the B rows through L1.  Row slices are split across workgroups (split-K) and
// combined with float atomics (256 contiguous bytes per wave instruction, the full-rate shape).
struct GemmTN { const float* A; int lda; int No; const float* B; int ldb; int Ni; float* C; int ldc; float* bias; int64_t M; int64_t rows_per_wg; };

#define GT_SLAB 32                     // rows per LDS slab = 16 k-steps
// One workgroup (8 waves) owns a 256(o) x 256(i) block of C for its row slice; wave w owns the 32 x 256 strip
// o in [32w, 32w+32) as 8 accumulator tiles.  A and B slabs of 32 rows are staged through LDS (coalesced
// 4-byte loads with column/row masks, so unaligned narrow operands such as x[:, :63] with ld 90 need no
// special case), double buffered: the next slab is fetched into registers while the current one feeds the
// MFMAs, then written to the other buffer behind one barrier per slab.
__global__ void __launch_bounds__(512, 2) gemm_tn_kernel(GemmTN P) {
    __shared__ float As[2][GT_SLAB][256];
    __shared__ float Bs[2][GT_SLAB][256];
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, hp = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int o0 = 32 * wv;
    const bool strip = o0 < P.No;                    // waves without a strip still load and synchronise
    const int i0 = 256 * blockIdx.y;
    const int nb = min(8, (P.Ni - i0 + 31) / 32);
    const int64_t m0 = (int64_t)blockIdx.x * P.rows_per_wg;
    const int64_t m1 = min(P.M, m0 + P.rows_per_wg);
    f32x16 acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    float bsum = 0.f;
    // slab element e = t + 512 k  ->  row e / 256, column e % 256 (consecutive threads, consecutive columns)
    const int lcol = t & 255, lrow0 = t >> 8;        // rows lrow0 + 2k
    const bool a_col_ok = lcol < P.No, b_col_ok = (i0 + lcol) < P.Ni;
    float ra[16], rb[16];
    auto fetch = [&](int64_t m) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t row = m + lrow0 + 2 * k;
            const bool ok = row < m1;
            ra[k] = (ok && a_col_ok) ? P.A[row * P.lda + lcol] : 0.f;
            rb[k] = (ok && b_col_ok) ? P.B[row * P.ldb + i0 + lcol] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { As[buf][lrow0 + 2 * k][lcol] = ra[k]; Bs[buf][lrow0 + 2 * k][lcol] = rb[k]; }
    };
    fetch(m0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int64_t m = m0; m < m1; m += GT_SLAB) {
        const bool more = (m + GT_SLAB) < m1;
        if (more) fetch(m + GT_SLAB);                // in flight while this slab is consumed
        if (strip) {
#pragma unroll 4
            for (int s = 0; s < GT_SLAB / 2; ++s) {
                const float a = As[buf][2 * s + hp][o0 + i];
                bsum += a;
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    if (b < nb) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[buf][2 * s + hp][32 * b + i], acc[b], 0, 0, 0);
            }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (!strip) return;
    // C/D map: register r of lane (j = i, h = hp) is row o0 + frow(r,h), column i0 + 32b + j
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        if (b >= nb) continue;
        const int col = i0 + 32 * b + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + sw_frow(r, hp);
            if (o < P.No && col < P.Ni) atomicAdd(P.C + (size_t)o * P.ldc + col, acc[b][r]);
        }
    }
    if (P.bias && blockIdx.y == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (hp == 0 && (o0 + i) < P.No) atomicAdd(P.bias + o0 + i, bsum);
    }
}

extern "C" int swnerf_gemm_tn(const float* A, int lda, int No, const float* B, int ldb, int Ni, int64_t M,
                              float* C, int ldc, float* bias, void* stream) {
    if (M == 0) return 0;
    if (!A || !B || !C || M < 0 || No < 1 || No > 256 || Ni < 1 || lda < No || ldb < Ni || ldc < Ni)
        return sw_fail(SWNERF_E_ARG, "gemm_tn: bad arguments (M=%lld No=%d Ni=%d lda=%d ldb=%d ldc=%d)", (long long)M, No, Ni, lda, ldb, ldc);
    GemmTN P;
    P.A = A; P.lda = lda; P.No = No; P.B = B; P.ldb = ldb; P.Ni = Ni; P.C = C; P.ldc = ldc; P.bias = bias; P.M = M;
    // split the rows over ~2 workgroups per CU, at least 256 rows each (whole slabs)
    int64_t nwg = (M + 255) / 256;
    if (nwg > 512) nwg = 512;
    int64_t rows = (M + nwg - 1) / nwg;
    rows = (rows + GT_SLAB - 1) / GT_SLAB * GT_SLAB;
    P.rows_per_wg = rows;
    nwg = (M + rows - 1) / rows;
    const dim3 grid((unsigned)nwg, (unsigned)((Ni + 255) / 256)), block(512);
    hipLaunchKernelGGL(gemm_tn_kernel, grid, block, 0, (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "gemm_tn launch");
}
