// backward_kernels.hip - gradients of the render path.
//
//  * raw2outputs backward: d(raw) from d(rgb_map, disp_map, acc_map, depth_map, weights)   (ray.py:155-198)
//  * TN GEMM  dW[o][i] += sum_m A[m][o] * B[m][i]  (+ column sums for the bias gradient): the weight
//    gradients of every Linear layer of the 8x256 MLP, v_mfma_f32_32x32x2_f32, K = rows
// (the register-resident dX chain lives in render_kernels.hip next to the forward it mirrors)
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "swnerf_common.h"
#include "lds_dma.h"
#include "wave_dpp.h"
#include "host_util.h"
#include <type_traits>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// raw2outputs backward, one wave per ray.  With c = sigmoid(rgb), e = exp(-relu(sigma)*dist),
// a = 1-e, p = 1-a+1e-10, T_i = prod_{j<i} p_j, w = a*T:
//   G_i   = dL/dw_i = g_rgb.c_i + gA + gD*z_i + g_w_i
//   dL/da_i = G_i*T_i - (sum_{k>i} G_k*w_k)/p_i ,   da/dsigma = dist*e*[sigma>0]
//   dL/drgb_i = w_i * g_rgb * c_i*(1-c_i)
// gA folds d(acc_map), the white-background term (rgb_map += 1-acc) and disp = 1/max(1e-10, D/A);
// gD folds d(depth_map) and disp.  Prefix products and suffix sums run in double like the forward.
// Both scans run on the DPP path (wave_dpp.h; the kernel is VALU-issue bound).  The suffix sums of pass 2 become PREFIX sums
// over lanes by handing the samples of a 64-sample chunk to the lanes in REVERSE order (lane L owns sample 64 ch + 63 - L): the
// loads stay one contiguous 1 KiB / 256 B block per wave instruction.  T and w of pass 1 wait in the wave's 2 x S floats of LDS.
#define R2B_SMAX 1024
__global__ void __launch_bounds__(256) raw2outputs_bwd_kernel(const float* raw, const float* zv, const float* rd, const float* noise,
                                                              int64_t N, int S, int white, const float* g_rgb, const float* g_disp,
                                                              const float* g_acc, const float* g_depth, const float* g_w, float* d_raw, int Sp) {
    extern __shared__ __attribute__((aligned(16))) float r2b_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
    if (ray >= N) return;
    float* T_ = r2b_lds + wv * 2 * Sp; float* W_ = T_ + Sp;
    const float dx = rd[ray * 3], dy = rd[ray * 3 + 1], dz = rd[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const float gr = g_rgb ? g_rgb[ray * 3] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    const float* zr = zv + ray * S;
    const float4* rr = reinterpret_cast<const float4*>(raw) + ray * S;
    const float* nr = noise ? noise + ray * S : nullptr;
    // pass 1: forward recompute of T, w; accumulate acc and depth for the disparity term
    double Tc = 1.0;
    float pa = 0.f, pd = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        const float z = zr[sc];
        const float z_edge = (lane == 63 && s + 1 < S) ? zr[s + 1] : 0.f;
        const float zn = wave_from_above_f32(z, z_edge);          // a cross-lane read: never under a lane-dependent branch
        float dist = (s + 1 < S) ? (zn - z) : 1e10f;
        dist *= dnorm;
        float sg = rr[sc].w;
        if (nr) sg += nr[sc];
        float alpha = 1.f - expf(-fmaxf(sg, 0.f) * dist);
        if (!live) alpha = 0.f;
        const double ps = wave_incl_prod_f64((double)(1.f - alpha + 1e-10f));
        const float T = (float)(Tc * wave_from_below_f64(ps, 1.0));
        Tc *= wave_last_f64(ps);
        const float w = alpha * T;
        if (live) { T_[s] = T; W_[s] = w; }
        pa += w; pd += w * z;
    }
    pa = __shfl(wave_sum_to_last_f32(pa), 63, 64); pd = __shfl(wave_sum_to_last_f32(pd), 63, 64);
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
        const int s = ch * 64 + 63 - lane;               // reversed: lane 0 owns the chunk's last sample
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        const float4 r4 = rr[sc];
        const float z = zr[sc];
        const float c0 = 1.f / (1.f + expf(-r4.x)), c1 = 1.f / (1.f + expf(-r4.y)), c2 = 1.f / (1.f + expf(-r4.z));
        const float w = live ? W_[sc] : 0.f, T = live ? T_[sc] : 0.f;
        float G = gr * c0 + gg * c1 + gb * c2 + gA + gD * z;
        if (g_w) G += g_w[ray * S + sc];
        // inclusive sum of G*w over the lanes <= this one = the samples >= s of the chunk, in double
        const double v = live ? (double)G * (double)w : 0.0;
        const double incl = wave_incl_sum_f64(v);
        const double R = carry + wave_from_below_f64(incl, 0.0);     // everything strictly behind sample s
        carry += wave_last_f64(incl);
        const float z_edge = (lane == 0 && s + 1 < S) ? zr[s + 1] : 0.f;      // sample s+1 sits one lane BELOW
        const float zn = dpp_f32<SW_DPP_WAVE_SHR1>(z_edge, z);    // (cross-lane: outside the lane-dependent select)
        float dist = (s + 1 < S) ? (zn - z) : 1e10f;
        dist *= dnorm;
        float sg = r4.w;
        if (nr) sg += nr[sc];
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
    const int Sp = (S + 63) & ~63;
    hipLaunchKernelGGL(raw2outputs_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), (size_t)4 * 2 * Sp * sizeof(float), (hipStream_t)stream,
                       raw, z_vals, rays_d, noise, N, S, white_bkgd, g_rgb, g_disp, g_acc, g_depth, g_weights, d_raw, Sp);
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
// One workgroup (8 waves) owns a 128(o) x 256(i) block of C for its row slice; wave w owns the 32 x 128 strip
// o in [32(w&3), +32), i in [128(w>>2), +128) as 4 accumulator tiles.  A (32 x 128) and B (32 x 256) slabs are
// staged through 48 KB of LDS by LDS-DMA, SINGLE buffered: DMA -> vmcnt(0) -> barrier -> MFMA -> barrier.  Overlap of
// memory and matrix work comes from 2 co-resident workgroups per CU (one loads while another computes); the
// 128-register budget that allows them is why nothing may be staged through registers (round 1 did, and spilled).
// VA / VB: that operand is 16-byte aligned with a column count that is a multiple of 4 -> 16-byte DMAs; otherwise
// 4-byte DMAs.
template <bool VA, bool VB>
__global__ void __launch_bounds__(512, 4) gemm_tn_kernel(GemmTN P) {
    __shared__ __attribute__((aligned(16))) float As[GT_SLAB][128];
    __shared__ __attribute__((aligned(16))) float Bs[GT_SLAB][256];
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, hp = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    // grid.y = (256-column blocks of C) x (halves of its up to 256 rows: 2 only when No > 128 - the launch's formula).
    // (Round 2 decoded y as if the factor were always 2: with No <= 128 and Ni > 256 - views_linears.0 of a W = 256 net on
    // the generic path, 128 x 283 - block y = 1 then read A 128 columns to the right of its rows and the columns of C past
    // 256 were never accumulated; found in round 3 by a fault on operands whose allocation ended with their last row.)
    const int nsplit = P.No > 128 ? 2 : 1;
    const int obase = 128 * (blockIdx.y % nsplit);   // which half of the (up to) 256 output rows
    const int i0 = 256 * (blockIdx.y / nsplit);
    const int o0 = 32 * (wv & 3), ih = 128 * (wv >> 2);
    const int nb = max(0, min(4, (P.Ni - i0 - ih + 31) / 32));
    const bool strip = (obase + o0) < P.No && nb > 0;
    const int64_t m0 = (int64_t)blockIdx.x * P.rows_per_wg;
    const int mlen = (int)(min(P.M, m0 + P.rows_per_wg) - m0);
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    float bsum = 0.f;
    const float* Ab = P.A + m0 * P.lda + obase;
    const float* Bb = P.B + m0 * P.ldb + i0;
    const unsigned as_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&As[0][0]);
    const unsigned bs_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&Bs[0][0]);
    for (int mrel = 0; mrel < mlen; mrel += GT_SLAB) {
        // ---- stage the slab by LDS-DMA: no staging registers at all (with register staging this kernel spilled
        // 56-176 B/lane at its 128-register budget, and scratch traffic shares vmcnt with everything else).  An operand
        // whose base, leading dimension and column count allow 16-byte accesses (VA / VB) moves 1 KiB per instruction
        // (`global_load_lds_dwordx4`), any other (x[:, :63] with ld 90, d_out[:, 3] with ld 4) 256 B per instruction
        // (`global_load_lds_dword`, one float per lane).  Nothing can be zero-filled on the way, so rows past the slice
        // and columns past the operand are CLAMPED to valid addresses instead: a clamped column only feeds C entries
        // that are never written, a clamped row is masked on the A side in the MFMA loop below.
        const unsigned rlast = (unsigned)(mlen - 1 - mrel);
        if (VA) {
            const unsigned col = (obase + 4 * i < P.No) ? 4u * i : 0u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {            // A: 32 rows x 512 B = 16 DMAs of two rows, 2 per wave
                const int r0 = 2 * (wv * 2 + q);
                ws_dma(reinterpret_cast<const char*>(Ab), ((mrel + min((unsigned)(r0 + hp), rlast)) * (unsigned)P.lda + col) * 4u,
                       as_addr + (unsigned)r0 * 512u);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {            // 4096 floats = 64 DMAs of 64 floats (half a row), 8 per wave
                const int f0 = 512 * k + 64 * wv, row = f0 >> 7, c = (f0 & 127) + lane;
                lds_dma_dword(reinterpret_cast<const char*>(Ab),
                              ((mrel + min((unsigned)row, rlast)) * (unsigned)P.lda + ((obase + c < P.No) ? c : 0)) * 4u, as_addr + (unsigned)f0 * 4u);
            }
        }
        if (VB) {
            const unsigned col = (i0 + 4 * lane < P.Ni) ? 4u * lane : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // B: 32 rows x 1 KiB = 32 DMAs, 4 per wave
                const int row = wv * 4 + q;
                ws_dma(reinterpret_cast<const char*>(Bb), ((mrel + min((unsigned)row, rlast)) * (unsigned)P.ldb + col) * 4u,
                       bs_addr + (unsigned)row * 1024u);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {           // 8192 floats = 128 DMAs of 64 floats (a quarter row), 16 per wave
                const int f0 = 512 * k + 64 * wv, row = f0 >> 8, c = (f0 & 255) + lane;
                lds_dma_dword(reinterpret_cast<const char*>(Bb),
                              ((mrel + min((unsigned)row, rlast)) * (unsigned)P.ldb + ((i0 + c < P.Ni) ? c : 0)) * 4u, bs_addr + (unsigned)f0 * 4u);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (strip) {
            if (nb == 4) {                           // full-width strip: batched LDS reads, branch-free MFMA groups
#pragma unroll 2
                for (int s = 0; s < GT_SLAB / 2; ++s) {
                    float a = As[2 * s + hp][o0 + i];
                    a = ((unsigned)(2 * s + hp) <= rlast) ? a : 0.f;              // rows past the slice (clamped copies)
                    float bv[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) bv[b] = Bs[2 * s + hp][ih + 32 * b + i];
                    bsum += a;
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[b], acc[b], 0, 0, 0);
                }
            } else {
#pragma unroll 2
                for (int s = 0; s < GT_SLAB / 2; ++s) {
                    float a = As[2 * s + hp][o0 + i];
                    a = ((unsigned)(2 * s + hp) <= rlast) ? a : 0.f;
                    bsum += a;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (b < nb) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[2 * s + hp][ih + 32 * b + i], acc[b], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (!strip) return;
    // C/D map: register r of lane (j = i, h = hp) is row obase + o0 + frow(r,h), column i0 + ih + 32b + j.
    // The lane index is taken afresh from mbcnt here: carried over from the prologue it stays live across the slab
    // loop, and at this kernel's 128-register budget that was the one value hipcc still spilled.
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int ie = lane_e & 31, he = lane_e >> 5;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b >= nb) continue;
        const int col = i0 + ih + 32 * b + ie;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = obase + o0 + sw_frow(r, he);
            if (o < P.No && col < P.Ni) atomicAdd(P.C + (size_t)o * P.ldc + col, acc[b][r]);
        }
    }
    if (P.bias && i0 == 0 && ih == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (he == 0 && (obase + o0 + ie) < P.No) atomicAdd(P.bias + obase + o0 + ie, bsum);
    }
}

// ---------------------------------------------------------------------------------------------
// The 256 x 256 case (every trunk layer, feature_linear: 16 of the 28 GEMMs of a training step and ~3/4 of their
// time): one workgroup of 16 waves owns the WHOLE 256 x 256 block of C for its row slice, so A and B are each read
// from HBM exactly once, and the 32-row slabs are DOUBLE buffered in 128 KB of LDS, filled by LDS-DMA (no staging
// registers for hipcc to sink or spill; the next slab is in flight while this one feeds the MFMAs).  One barrier per
// slab.  Wave w owns the 64 x 64 block o in [64(w&3), +64), i in [64(w>>2), +64) as 2 x 2 accumulator tiles:
// 4 LDS reads per 4 MFMAs.
#define GD_SLAB 32
#ifndef GD_RIDER_UNR
#define GD_RIDER_UNR 4                    // k-pairs unrolled in an item with a rider
#endif
#define GD_BUF_FLOATS (2 * GD_SLAB * 256)            // A slab then B slab
#define GD_B2_FLOATS (GD_SLAB * 64)                  // optional second B operand, <= 64 columns
#define GD_A2_FLOATS (GD_SLAB * 32)                  // optional second A operand, <= 32 columns (zero padded)
// Two optional riders on the same pass (each one extra accumulator tile per wave):
//   B2 [M, Ni2 <= 64]:  C2[256, Ni2] += A^T . B2   - the gamma(x) columns of the skip layer, whose dW shares A = d pre_5
//                       with the 256-wide part (cat[gamma(x), h4], model.py:45-46)
//   A2 [M, No2 <= 32]:  C3[No2, 256] += A2^T . B, bias3 += column sums of A2   - alpha_linear, which shares B = h7 with
//                       feature_linear
// Their slabs are small and unaligned (ld 90, ld 4): staged through registers by plain loads issued at the top of the
// compute phase and written to LDS behind it, double buffered like the DMA slabs.
struct GemmFused {
    GemmTN g;
    const float* B2; int ldb2; int Ni2; float* C2; int ldc2;
    const float* A2; int lda2; int No2; float* C3; int ldc3; float* bias3;
};

template <bool HB2, bool HA2>
__device__ __forceinline__ void gemm_dma_body(const GemmFused& F, const int slice) {
    const GemmTN& P = F.g;
    extern __shared__ __attribute__((aligned(16))) float gd_lds[];           // [2][GD_BUF_FLOATS] [2][B2] [2][A2]
    float* b2s = gd_lds + 2 * GD_BUF_FLOATS;
    float* a2s = b2s + (HB2 ? 2 * GD_B2_FLOATS : 0);
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, hp = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int o0 = 64 * (w & 3), i0 = 64 * (w >> 2);
    const int64_t m0 = (int64_t)slice * P.rows_per_wg;
    const int mlen = (int)(min(P.M, m0 + P.rows_per_wg) - m0);
    const int nslab = (mlen + GD_SLAB - 1) / GD_SLAB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)gd_lds);
    const unsigned voff = (unsigned)lane * 16u;
    // slab s -> buffer s&1: 64 rows of 1 KiB (32 of A, 32 of B), 4 per wave.  The row pointers advance by one slab
    // per call (a 64-bit add each; recomputing them costs ~80 dependent scalar instructions per slab, which all
    // 16 waves execute at the same moment right after the barrier, with the matrix pipe idle).  Rows past the slice
    // (last slab only) are clamped to its last row - never out of bounds - and zeroed on the A side when read.
    const char* cur[4];
    int64_t step[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int id = w * 4 + q, row = id & 31;
        cur[q] = reinterpret_cast<const char*>(id < 32 ? P.A + (m0 + row) * P.lda : P.B + (m0 + row) * P.ldb);
        step[q] = (int64_t)GD_SLAB * 4 * (id < 32 ? P.lda : P.ldb);
    }
    auto issue = [&](int sl) {
        const bool full = (sl + 1) * GD_SLAB <= mlen;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = w * 4 + q, row = id & 31;
            const char* g = cur[q];
            if (!full) {
                const int64_t r = m0 + min(sl * GD_SLAB + row, mlen - 1);
                g = reinterpret_cast<const char*>(id < 32 ? P.A + r * P.lda : P.B + r * P.ldb);
            }
            ws_dma(g, voff, lds0 + (unsigned)((sl & 1) * GD_BUF_FLOATS * 4 + id * 1024));
            cur[q] += step[q];
        }
    };
    // rider slabs: element e of the B2 slab is (row e>>6, col e&63), of the A2 slab (row e>>5, col e&31); loads are
    // unconditional (clamped), the zero fill happens at the LDS write
    float rb2[2] = {0.f, 0.f}, ra2 = 0.f;
    auto rider_load = [&](int sl) {
        if (HB2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = t + 1024 * k, row = e >> 6, col = e & 63;
                rb2[k] = F.B2[(m0 + min(sl * GD_SLAB + row, mlen - 1)) * F.ldb2 + (col < F.Ni2 ? col : 0)];
            }
        }
        if (HA2) {
            const int row = t >> 5, col = t & 31;
            ra2 = F.A2[(m0 + min(sl * GD_SLAB + row, mlen - 1)) * F.lda2 + (col < F.No2 ? col : 0)];
        }
    };
    auto rider_store = [&](int sl) {
        if (HB2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = t + 1024 * k, col = e & 63;
                b2s[(sl & 1) * GD_B2_FLOATS + e] = (col < F.Ni2) ? rb2[k] : 0.f;
            }
        }
        if (HA2) {
            const int row = t >> 5, col = t & 31;
            a2s[(sl & 1) * GD_A2_FLOATS + t] = (col < F.No2 && sl * GD_SLAB + row < mlen) ? ra2 : 0.f;
        }
    };
    f32x16 acc[4], accb, acca;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        accb[r] = 0.f; acca[r] = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b][r] = 0.f;
    }
    f32x2 bs01 = {0.f, 0.f};                                 // column sums of this lane's two A columns
    float bs3 = 0.f;
    const bool do_bias = P.bias != nullptr && i0 == 0;       // (w and P.bias are wave-uniform: a scalar branch)
    const int ot2 = 32 * (w & 7), it2 = 32 * (w >> 3);       // B2 rider: this wave's 32 x 32 tile of C2
    issue(0);
    if (HB2 || HA2) { rider_load(0); rider_store(0); }
    // -DGEMM_EXP_* (tools/experiments/gemm/build.sh; timing experiments, WRONG results, never the shipped library):
    // NOBARRIER no slab barrier | NODMA only the first two slabs are ever fetched | NOVALU no row masks / bias sums | NOEPI no atomics
    // The WHOLE slab loop once per bias role (the test in front of it, not inside: with two copies of the unrolled steps inside the
    // loop hipcc keeps accumulator tiles alive across both and spills 150+ registers at this kernel's 128-register budget - as in
    // narrow5_kernel).  Both copies execute the same barriers.
    auto slabs = [&](auto bias_) {
    constexpr bool BIAS = decltype(bias_)::value;
#pragma nounroll
    for (int sl = 0; sl < nslab; ++sl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of slab sl has landed ...
#ifndef GEMM_EXP_NOBARRIER
        __syncthreads();                                      // ... everyone's has; and everyone is done with slab sl-1
#endif
        float* Abw = gd_lds + (sl & 1) * GD_BUF_FLOATS;
        const int valid = mlen - sl * GD_SLAB;                // rows of this slab inside the slice (>= 32: all)
        if (valid < GD_SLAB) {
            // the slice's last, partial slab (once per workgroup): its rows past the slice hold clamped copies of the last row -
            // zero them on the A side HERE instead of masking every operand of every k-pair (two selects per 4 MFMAs on all 16
            // waves cost 3-4 % of the whole launch: profiles/r04/gemm_exp.md)
            for (int e = t; e < (GD_SLAB - valid) * 256; e += 1024) Abw[valid * 256 + e] = 0.f;
            __syncthreads();
        }
        const float* Ab = Abw;
        // The wave's 64 x 64 block of C is FOUR INTERLEAVED tiles - rows o0 + 2m + {0,1} x columns i0 + 2n + {0,1} - so that lane i's two
        // A operands of a k-pair (columns o0 + 2i, o0 + 2i + 1 of one row) and its two B operands are ONE ds_read_b64 each, offset in
        // the instruction: no address arithmetic sits between the MFMAs (tiles of 32 adjacent columns are 128 B apart - a ds_read2_b32
        // reaches 1 KiB, i.e. two v_add_u32 per k-pair on all 16 waves: 7 % of the launch, profiles/r04/gemm_exp.md), and the bias column
        // sums are one v_pk_add_f32.  Operands of k-pair s+1 are read while the MFMAs of k-pair s run; the first reads go out BEFORE the
        // next slab's DMA is issued, so that its scalar address work hides under their LDS latency.
        const f32x2* A2p = reinterpret_cast<const f32x2*>(Ab + o0 + 2 * i + hp * 256);
        const f32x2* B2p = reinterpret_cast<const f32x2*>(Ab + GD_SLAB * 256 + i0 + 2 * i + hp * 256);
        f32x2 a = A2p[0], b = B2p[0];
        // rider operands: per-slab lane bases as opaque float indices into the LDS array (the B2 / A2 slabs lie beyond the 64 KiB an
        // instruction offset reaches: left to itself hipcc re-adds the 128 KiB constant before every read)
        int e1o = (int)(b2s - gd_lds) + (sl & 1) * GD_B2_FLOATS + hp * 64 + it2 + i;
        int f0o = (int)(a2s - gd_lds) + (sl & 1) * GD_A2_FLOATS + hp * 32 + i;
        if (HB2) asm("" : "+v"(e1o));
        if (HA2) asm("" : "+v"(f0o));
        __builtin_amdgcn_sched_barrier(0);
#ifdef GEMM_EXP_NODMA
        if (sl + 1 < 2) {
#else
        if (sl + 1 < nslab) {
#endif
            issue(sl + 1);
            if (HB2 || HA2) rider_load(sl + 1);
        }
        {
            constexpr int UNR = (HB2 || HA2) ? GD_RIDER_UNR : GD_SLAB / 2;    // a rider's extra tile leaves fewer registers for the unroll
#pragma unroll UNR
            for (int s = 0; s < GD_SLAB / 2; ++s) {
                const int row = 2 * s + hp;
                const f32x2 c = a, d = b;                     // (rows past the slice are zero on the A side: see above)
                float e0 = 0.f, e1 = 0.f, f0 = 0.f, f1 = 0.f;
                if (HB2) {                                    // A columns of this wave's C2 tile x B2 columns
                    e0 = Ab[row * 256 + ot2 + i];
                    e1 = gd_lds[e1o + 2 * s * 64];
                }
                if (HA2) {                                    // A2 columns (zero padded) x B columns 32(w&7).. (waves 8..15
                    f0 = gd_lds[f0o + 2 * s * 32];                         // duplicate 0..7 rather than branch; only 0..7 write)
                    f1 = Ab[GD_SLAB * 256 + row * 256 + 32 * (w & 7) + i];
                }
                {                                             // (the last k-pair reads the slab's last two rows again: harmless, unused)
                    const int nr = min(2 * s + 2, GD_SLAB - 2) * 128;
                    a = A2p[nr]; b = B2p[nr];
                }
                __builtin_amdgcn_sched_barrier(0);
#ifndef GEMM_EXP_NOVALU
                if (BIAS) asm("v_pk_add_f32 %0, %0, %1" : "+v"(bs01) : "v"(c));    // (hipcc splits a two-float vector add into two v_add_f32)
#endif
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c[0], d[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c[0], d[1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(c[1], d[0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(c[1], d[1], acc[3], 0, 0, 0);
                if (HB2) accb = __builtin_amdgcn_mfma_f32_32x32x2f32(e0, e1, accb, 0, 0, 0);
                if (HA2) { acca = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, f1, acca, 0, 0, 0); bs3 += f0; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if ((HB2 || HA2) && sl + 1 < nslab) rider_store(sl + 1);   // visible after the next barrier
    }
    };
    if (do_bias) slabs(std::true_type{}); else slabs(std::false_type{});   // wave-uniform: only the four waves of the first column block own bias entries
#ifdef GEMM_EXP_NOEPI
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + accb[0] + acca[0] + bs01[0] + bs01[1] + bs3 != 12345.678f) return;
#endif
    // C/D map: register r of lane (j = i, h = hp) of tile (oa, ib) is row o0 + 2 frow(r,h) + oa, column i0 + 2 j + ib (interleaved tiles)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int col = i0 + 2 * i + (b & 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + 2 * sw_frow(r, hp) + (b >> 1);
            atomicAdd(P.C + (size_t)o * P.ldc + col, acc[b][r]);
        }
    }
    if (do_bias) {
        float bs0 = bs01[0], bs1 = bs01[1];
        bs0 += __shfl_xor(bs0, 32, 64); bs1 += __shfl_xor(bs1, 32, 64);
        if (hp == 0) { atomicAdd(P.bias + o0 + 2 * i, bs0); atomicAdd(P.bias + o0 + 2 * i + 1, bs1); }
    }
    if (HB2 && it2 + i < F.Ni2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(F.C2 + (size_t)(ot2 + sw_frow(r, hp)) * F.ldc2 + it2 + i, accb[r]);
    }
    if (HA2 && w < 8) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (sw_frow(r, hp) < F.No2) atomicAdd(F.C3 + (size_t)sw_frow(r, hp) * F.ldc3 + 32 * w + i, acca[r]);
        if (F.bias3 && w == 0) {
            bs3 += __shfl_xor(bs3, 32, 64);
            if (hp == 0 && i < F.No2) atomicAdd(F.bias3 + i, bs3);
        }
    }
}

template <bool HB2, bool HA2>
__global__ void __launch_bounds__(1024) gemm_tn_dma_kernel(GemmFused F) { gemm_dma_body<HB2, HA2>(F, (int)blockIdx.x); }

// SEVERAL such GEMMs of one row chunk as ONE launch (swnerf_gemm_tn_group): the weight-gradient GEMMs of a chunk are
// independent, and every launch costs ~70 us that the matrix pipe idles through (ramp, and an epilogue of 64 K float atomics
// per workgroup that all workgroups reach at the same moment) - 15-24 % of a 393 216-row launch.  Here the ~256 workgroups
// of ONE launch are dealt out over the items in proportion to their work (an item with a rider does 5 MFMAs per 4), each
// covering a longer row slice of its item: one ramp and one epilogue per chunk instead of one per layer.
#define GG_MAX 16
struct GemmGroup { int n; int wg0[GG_MAX + 1]; GemmFused it[GG_MAX]; };      // wg0: first workgroup of item k (prefix sums)
__global__ void __launch_bounds__(1024) gemm_tn_dma_group_kernel(GemmGroup G) {
    int k = 0;
    while (k + 1 < G.n && (int)blockIdx.x >= G.wg0[k + 1]) ++k;               // wave-uniform: scalar
    const GemmFused& F = G.it[k];
    const int slice = (int)blockIdx.x - G.wg0[k];
    if (F.B2) gemm_dma_body<true, false>(F, slice);
    else if (F.A2) gemm_dma_body<false, true>(F, slice);
    else gemm_dma_body<false, false>(F, slice);
}

// ---------------------------------------------------------------------------------------------
// The same plan for the SKINNY weight-gradient GEMMs of a training step - pts_linears.0 (256 x 64 slots of gamma(x)),
// views_linears.0 (128 x 256 and 128 x 32), rgb_linear (4 x 128), the deformation net's gamma(t) columns and _time_out:
// 6 % of the FLOPs that took 16 % of the GEMM time on the single-buffered narrow kernel above (2 TB/s).  One 16-wave
// workgroup per CU-sized row slice, 32-row slabs of A and B double buffered in LDS by LDS-DMA (1 KiB row pitch whatever
// the operand's width: lanes past its last column re-read its first 16 bytes, never another row), one barrier per slab.
// The WO x WI wave grid covers C with TO x TI accumulator tiles per wave; waves beyond the grid only help with the DMA.
// These shapes are HBM bound: what matters is that a whole slab per CU is always in flight.
template <int TO, int TI, int WO, int WI>
__global__ void __launch_bounds__(1024) gemm_tn_tiled_kernel(GemmTN P) {
    static_assert(WO * WI <= 16 && 32 * TO * WO <= 256 && 32 * TI * WI <= 256, "wave grid must fit the 16-wave workgroup and the 256-column slabs");
    extern __shared__ __attribute__((aligned(16))) float gd_lds[];           // [2][GD_BUF_FLOATS]
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, hp = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool active = w < WO * WI;
    const int o0 = 32 * TO * (w % WO), i0 = 32 * TI * (w / WO);
    const int64_t m0 = (int64_t)blockIdx.x * P.rows_per_wg;
    const int mlen = (int)(min(P.M, m0 + P.rows_per_wg) - m0);
    const int nslab = (mlen + GD_SLAB - 1) / GD_SLAB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)gd_lds);
    const unsigned voff_a = (4 * lane < P.No) ? (unsigned)lane * 16u : 0u;
    const unsigned voff_b = (4 * lane < P.Ni) ? (unsigned)lane * 16u : 0u;
    const char* cur[4];
    int64_t step[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int id = w * 4 + q, row = id & 31;
        cur[q] = reinterpret_cast<const char*>(id < 32 ? P.A + (m0 + row) * P.lda : P.B + (m0 + row) * P.ldb);
        step[q] = (int64_t)GD_SLAB * 4 * (id < 32 ? P.lda : P.ldb);
    }
    auto issue = [&](int sl) {
        const bool full = (sl + 1) * GD_SLAB <= mlen;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int id = w * 4 + q, row = id & 31;
            const char* g = cur[q];
            if (!full) {
                const int64_t r = m0 + min(sl * GD_SLAB + row, mlen - 1);
                g = reinterpret_cast<const char*>(id < 32 ? P.A + r * P.lda : P.B + r * P.ldb);
            }
            ws_dma(g, id < 32 ? voff_a : voff_b, lds0 + (unsigned)((sl & 1) * GD_BUF_FLOATS * 4 + id * 1024));
            cur[q] += step[q];
        }
    };
    f32x16 acc[TO * TI];
    float bs[TO];
#pragma unroll
    for (int k = 0; k < TO * TI; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
    for (int a = 0; a < TO; ++a) bs[a] = 0.f;
    const bool do_bias = P.bias != nullptr && i0 == 0;       // (wave-uniform)
    issue(0);
#pragma nounroll
    for (int sl = 0; sl < nslab; ++sl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of slab sl has landed ...
        __syncthreads();                                      // ... everyone's has; and everyone is done with slab sl-1
        float* Abw = gd_lds + (sl & 1) * GD_BUF_FLOATS;
        const int valid = mlen - sl * GD_SLAB;                // rows of this slab inside the slice (>= 32: all)
        if (valid < GD_SLAB) {
            // the slice's last, partial slab (once per workgroup, ALL waves: also the ones that only help with the DMA): its rows past
            // the slice hold clamped copies of the last row - zero them on the A side instead of masking every operand of every k-pair
            for (int e = t; e < (GD_SLAB - valid) * 256; e += 1024) Abw[valid * 256 + e] = 0.f;
            __syncthreads();
        }
        if (sl + 1 < nslab) issue(sl + 1);
        if (!active) continue;
        const float* Ab = Abw;
        const float* As = Ab + o0 + i;
        const float* Bs = Ab + GD_SLAB * 256 + i0 + i;
        float a[TO], b[TI];
#pragma unroll
        for (int x = 0; x < TO; ++x) a[x] = As[hp * 256 + 32 * x];
#pragma unroll
        for (int x = 0; x < TI; ++x) b[x] = Bs[hp * 256 + 32 * x];
#pragma unroll
        for (int s = 0; s < GD_SLAB / 2; ++s) {
            const int row = 2 * s + hp;
            float c[TO], d[TI];
#pragma unroll
            for (int x = 0; x < TO; ++x) c[x] = a[x];         // (rows past the slice are zero on the A side: see above)
#pragma unroll
            for (int x = 0; x < TI; ++x) d[x] = b[x];
            const int nr = (min(row + 2, GD_SLAB - 1)) * 256;   // (the last iteration re-reads the slab's last rows: unused)
#pragma unroll
            for (int x = 0; x < TO; ++x) a[x] = As[nr + 32 * x];
#pragma unroll
            for (int x = 0; x < TI; ++x) b[x] = Bs[nr + 32 * x];
            __builtin_amdgcn_sched_barrier(0);
            if (do_bias) {                                    // only the waves of the first column block own bias entries
#pragma unroll
                for (int x = 0; x < TO; ++x) bs[x] += c[x];
            }
#pragma unroll
            for (int x = 0; x < TO; ++x)
#pragma unroll
                for (int y = 0; y < TI; ++y) acc[x * TI + y] = __builtin_amdgcn_mfma_f32_32x32x2f32(c[x], d[y], acc[x * TI + y], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!active) return;
    // C/D map: register r of lane (j = i, h = hp) of tile (x, y) is row o0 + 32 x + frow(r,h), column i0 + 32 y + j
#pragma unroll
    for (int x = 0; x < TO; ++x)
#pragma unroll
        for (int y = 0; y < TI; ++y) {
            const int col = i0 + 32 * y + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * x + sw_frow(r, hp);
                if (o < P.No && col < P.Ni) atomicAdd(P.C + (size_t)o * P.ldc + col, acc[x * TI + y][r]);
            }
        }
    if (do_bias) {
#pragma unroll
        for (int x = 0; x < TO; ++x) {
            const float v = bs[x] + __shfl_xor(bs[x], 32, 64);
            if (hp == 0 && o0 + 32 * x + i < P.No) atomicAdd(P.bias + o0 + 32 * x + i, v);
        }
    }
}

template <int TO, int TI, int WO, int WI>
static int gemm_tiled_launch(GemmTN P, void* stream) {
    int64_t nwg = 256;                                   // one workgroup per CU-sized row slice, whole slabs
    int64_t rows = ((P.M + nwg - 1) / nwg + GD_SLAB - 1) / GD_SLAB * GD_SLAB;
    nwg = (P.M + rows - 1) / rows;
    P.rows_per_wg = rows;
    hipLaunchKernelGGL((gemm_tn_tiled_kernel<TO, TI, WO, WI>), dim3((unsigned)nwg), dim3(1024), 2 * GD_BUF_FLOATS * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "gemm_tn (tiled) launch");
}

static int gemm_dma_launch(const GemmFused& F0, void* stream) {
    GemmFused F = F0;
    const int64_t M = F.g.M;
    int64_t nwg = 256;                                   // one workgroup per CU-sized row slice, whole slabs
    int64_t rows = ((M + nwg - 1) / nwg + GD_SLAB - 1) / GD_SLAB * GD_SLAB;
    nwg = (M + rows - 1) / rows;
    F.g.rows_per_wg = rows;
    const bool hb2 = F.B2 != nullptr, ha2 = F.A2 != nullptr;
    const size_t lds = (2 * GD_BUF_FLOATS + (hb2 ? 2 * GD_B2_FLOATS : 0) + (ha2 ? 2 * GD_A2_FLOATS : 0)) * sizeof(float);
    const dim3 grid((unsigned)nwg), block(1024);
    hipStream_t st = (hipStream_t)stream;
    if (hb2 && ha2) return sw_fail(SWNERF_E_ARG, "gemm_tn (dma): one rider per launch");
    else if (hb2) hipLaunchKernelGGL((gemm_tn_dma_kernel<true, false>), grid, block, lds, st, F);
    else if (ha2) hipLaunchKernelGGL((gemm_tn_dma_kernel<false, true>), grid, block, lds, st, F);
    else hipLaunchKernelGGL((gemm_tn_dma_kernel<false, false>), grid, block, lds, st, F);
    return sw_check(hipGetLastError(), "gemm_tn (dma) launch");
}

extern "C" int swnerf_gemm_tn(const float* A, int lda, int No, const float* B, int ldb, int Ni, int64_t M,
                              float* C, int ldc, float* bias, void* stream);

// The 256 x 256 GEMM with riders (see gemm_tn_dma_kernel).  Falls back to separate swnerf_gemm_tn calls when the
// main operands do not qualify for the DMA kernel (alignment, M < 4096).
extern "C" int swnerf_gemm_tn_fused(const float* A, int lda, const float* B, int ldb, int64_t M, float* C, int ldc, float* bias,
                                    const float* B2, int ldb2, int Ni2, float* C2, int ldc2,
                                    const float* A2, int lda2, int No2, float* C3, int ldc3, float* bias3, void* stream) {
    if (M == 0) return 0;
    if (!A || !B || !C || M < 0 || lda < 256 || ldb < 256 || ldc < 256)
        return sw_fail(SWNERF_E_ARG, "gemm_tn_fused: bad main operands (M=%lld lda=%d ldb=%d ldc=%d)", (long long)M, lda, ldb, ldc);
    if (B2 && (!C2 || Ni2 < 1 || Ni2 > 64 || ldb2 < Ni2 || ldc2 < Ni2)) return sw_fail(SWNERF_E_ARG, "gemm_tn_fused: bad B2 rider (Ni2=%d)", Ni2);
    if (A2 && (!C3 || No2 < 1 || No2 > 32 || lda2 < No2 || ldc3 < 256)) return sw_fail(SWNERF_E_ARG, "gemm_tn_fused: bad A2 rider (No2=%d)", No2);
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
    if (aligned && M >= 4096) {
        GemmFused F;
        F.g.A = A; F.g.lda = lda; F.g.No = 256; F.g.B = B; F.g.ldb = ldb; F.g.Ni = 256; F.g.C = C; F.g.ldc = ldc; F.g.bias = bias; F.g.M = M;
        F.B2 = B2; F.ldb2 = ldb2; F.Ni2 = Ni2; F.C2 = C2; F.ldc2 = ldc2;
        F.A2 = A2; F.lda2 = lda2; F.No2 = No2; F.C3 = C3; F.ldc3 = ldc3; F.bias3 = bias3;
        if (B2 && A2) {                                  // one rider per launch: A2 goes on its own
            F.A2 = nullptr;
            int rc2 = swnerf_gemm_tn(A2, lda2, No2, B, ldb, 256, M, C3, ldc3, bias3, stream);
            if (rc2) return rc2;
        }
        return gemm_dma_launch(F, stream);
    }
    int rc = swnerf_gemm_tn(A, lda, 256, B, ldb, 256, M, C, ldc, bias, stream);
    if (!rc && B2) rc = swnerf_gemm_tn(A, lda, 256, B2, ldb2, Ni2, M, C2, ldc2, nullptr, stream);
    if (!rc && A2) rc = swnerf_gemm_tn(A2, lda2, No2, B, ldb, 256, M, C3, ldc3, bias3, stream);
    return rc;
}

// Up to 16 of the 256 x 256 GEMMs (each with at most one rider) over the SAME M rows as one launch; see
// gemm_tn_dma_group_kernel.  Items that do not qualify for the DMA kernel (alignment, M < 4096, both riders) go through
// swnerf_gemm_tn_fused one by one.
extern "C" int swnerf_gemm_tn_group(const swnerf_gemm_item* items, int n_items, int64_t M, void* stream) {
    if (M == 0 || n_items == 0) return 0;
    if (!items || n_items < 0 || M < 0) return sw_fail(SWNERF_E_ARG, "gemm_tn_group: NULL items / negative count");
    GemmGroup G;
    G.n = 0;
    int weight[GG_MAX];
    bool any_b2 = false, any_a2 = false;
    auto single = [&](const swnerf_gemm_item& q) {
        return swnerf_gemm_tn_fused(q.A, q.lda, q.B, q.ldb, M, q.C, q.ldc, q.bias, q.B2, q.ldb2, q.Ni2, q.C2, q.ldc2,
                                    q.A2, q.lda2, q.No2, q.C3, q.ldc3, q.bias3, stream);
    };
    for (int k = 0; k < n_items; ++k) {
        const swnerf_gemm_item& q = items[k];
        if (!q.A || !q.B || !q.C || q.lda < 256 || q.ldb < 256 || q.ldc < 256)
            return sw_fail(SWNERF_E_ARG, "gemm_tn_group: item %d: bad main operands (lda=%d ldb=%d ldc=%d)", k, q.lda, q.ldb, q.ldc);
        if (q.B2 && (!q.C2 || q.Ni2 < 1 || q.Ni2 > 64 || q.ldb2 < q.Ni2 || q.ldc2 < q.Ni2)) return sw_fail(SWNERF_E_ARG, "gemm_tn_group: item %d: bad B2 rider", k);
        if (q.A2 && (!q.C3 || q.No2 < 1 || q.No2 > 32 || q.lda2 < q.No2 || q.ldc3 < 256)) return sw_fail(SWNERF_E_ARG, "gemm_tn_group: item %d: bad A2 rider", k);
        const bool aligned = (q.lda % 4 == 0) && (q.ldb % 4 == 0) && (((uintptr_t)q.A | (uintptr_t)q.B) % 16 == 0);
        if (!aligned || M < 4096 || (q.B2 && q.A2) || G.n == GG_MAX || getenv("SWNERF_GEMM_GROUP_OFF") != nullptr) {
            int rc = single(q);
            if (rc) return rc;
            continue;
        }
        GemmFused& F = G.it[G.n];
        F.g.A = q.A; F.g.lda = q.lda; F.g.No = 256; F.g.B = q.B; F.g.ldb = q.ldb; F.g.Ni = 256; F.g.C = q.C; F.g.ldc = q.ldc; F.g.bias = q.bias; F.g.M = M;
        F.B2 = q.B2; F.ldb2 = q.ldb2; F.Ni2 = q.Ni2; F.C2 = q.C2; F.ldc2 = q.ldc2;
        F.A2 = q.A2; F.lda2 = q.lda2; F.No2 = q.No2; F.C3 = q.C3; F.ldc3 = q.ldc3; F.bias3 = q.bias3;
        static const int rider_w = getenv("SWNERF_GG_RIDER_W") ? atoi(getenv("SWNERF_GG_RIDER_W")) : 12;
        static const int plain_w = getenv("SWNERF_GG_PLAIN_W") ? atoi(getenv("SWNERF_GG_PLAIN_W")) : 8;
        weight[G.n] = (q.B2 || q.A2) ? rider_w : plain_w;         // 5 MFMAs per 4 and a shorter unroll: 1.2-1.3x alone, 6 : 4 measured best in a group
        any_b2 |= q.B2 != nullptr; any_a2 |= q.A2 != nullptr;
        ++G.n;
    }
    if (G.n == 0) return 0;
    if (G.n == 1) {                                          // nothing to group
        GemmFused F = G.it[0];
        return gemm_dma_launch(F, stream);
    }
    // ~256 workgroups (one per CU: each needs >128 KB of LDS) dealt out in proportion to the items' work, whole 32-row slabs each
    int wsum = 0;
    for (int k = 0; k < G.n; ++k) wsum += weight[k];
    int total = 0;
    G.wg0[0] = 0;
    for (int k = 0; k < G.n; ++k) {
        int64_t nwg = (256 * (int64_t)weight[k]) / wsum;
        if (nwg < 1) nwg = 1;
        int64_t rows = ((M + nwg - 1) / nwg + GD_SLAB - 1) / GD_SLAB * GD_SLAB;
        nwg = (M + rows - 1) / rows;
        G.it[k].g.rows_per_wg = rows;
        total += (int)nwg;
        G.wg0[k + 1] = total;
    }
    const size_t lds = (2 * GD_BUF_FLOATS + (any_b2 ? 2 * GD_B2_FLOATS : 0) + (any_a2 ? 2 * GD_A2_FLOATS : 0)) * sizeof(float);
    // (a body without the B2 rider places the A2 slabs right behind the main buffers: any_b2's space is then simply unused)
    hipLaunchKernelGGL(gemm_tn_dma_group_kernel, dim3((unsigned)total), dim3(1024), lds, (hipStream_t)stream, G);
    return sw_check(hipGetLastError(), "gemm_tn_group launch");
}

extern "C" int swnerf_gemm_tn(const float* A, int lda, int No, const float* B, int ldb, int Ni, int64_t M,
                              float* C, int ldc, float* bias, void* stream) {
    if (M == 0) return 0;
    if (!A || !B || !C || M < 0 || No < 1 || No > 256 || Ni < 1 || lda < No || ldb < Ni || ldc < Ni)
        return sw_fail(SWNERF_E_ARG, "gemm_tn: bad arguments (M=%lld No=%d Ni=%d lda=%d ldb=%d ldc=%d)", (long long)M, No, Ni, lda, ldb, ldc);
    GemmTN P;
    P.A = A; P.lda = lda; P.No = No; P.B = B; P.ldb = ldb; P.Ni = Ni; P.C = C; P.ldc = ldc; P.bias = bias; P.M = M;
    const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) % 16 == 0);
    if (aligned && No == 256 && Ni == 256 && M >= 4096) {
        GemmFused F;
        F.g = P;
        F.B2 = nullptr; F.ldb2 = 0; F.Ni2 = 0; F.C2 = nullptr; F.ldc2 = 0;
        F.A2 = nullptr; F.lda2 = 0; F.No2 = 0; F.C3 = nullptr; F.ldc3 = 0; F.bias3 = nullptr;
        return gemm_dma_launch(F, stream);
    }
    // skinny shapes with 16-byte aligned operands: the double-buffered LDS-DMA kernel with the wave grid that covers C
    if (aligned && No % 4 == 0 && Ni % 4 == 0 && M >= 4096 && getenv("SWNERF_GEMM_NARROW_OLD") == nullptr) {
        if (No <= 32 && Ni <= 128) return gemm_tiled_launch<1, 1, 1, 4>(P, stream);     // rgb_linear 4 x 128
        if (No <= 32 && Ni <= 256) return gemm_tiled_launch<1, 2, 1, 4>(P, stream);     // _time_out 4 x 256
        if (No <= 128 && Ni <= 32) return gemm_tiled_launch<1, 1, 4, 1>(P, stream);     // views_linears.0, gamma(d) slots 128 x 32
        if (Ni <= 32) return gemm_tiled_launch<1, 1, 8, 1>(P, stream);                  // _time.0, gamma(t) slots 256 x 32
        if (Ni <= 64) return gemm_tiled_launch<1, 1, 8, 2>(P, stream);                  // pts_linears.0 / _time.0, gamma(x) slots 256 x 64
        // (views_linears.0 x feature, 128 x 256, is matrix bound rather than HBM bound and measured no faster on this
        // kernel's <1,2,4,4> grid - 292 us against 282 us at 393 216 rows - so it stays on the kernel below)
    }
    // split the rows over ~2 workgroups per CU, at least 256 rows each (whole slabs)
    int64_t nwg = (M + 255) / 256;
    if (nwg > 512) nwg = 512;
    int64_t rows = (M + nwg - 1) / nwg;
    rows = (rows + GT_SLAB - 1) / GT_SLAB * GT_SLAB;
    P.rows_per_wg = rows;
    nwg = (M + rows - 1) / rows;
    if (rows * (int64_t)(lda > ldb ? lda : ldb) >= (1LL << 30)) return sw_fail(SWNERF_E_UNSUPP, "gemm_tn: row slice too large for 32-bit byte offsets");
    const dim3 grid((unsigned)nwg, (unsigned)(((Ni + 255) / 256) * (No > 128 ? 2 : 1))), block(512);
    const bool va = (lda % 4 == 0) && (No % 4 == 0) && ((uintptr_t)A % 16 == 0);
    const bool vb = (ldb % 4 == 0) && (Ni % 4 == 0) && ((uintptr_t)B % 16 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (va && vb) hipLaunchKernelGGL((gemm_tn_kernel<true, true>), grid, block, 0, st, P);
    else if (va) hipLaunchKernelGGL((gemm_tn_kernel<true, false>), grid, block, 0, st, P);
    else if (vb) hipLaunchKernelGGL((gemm_tn_kernel<false, true>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((gemm_tn_kernel<false, false>), grid, block, 0, st, P);
    return sw_check(hipGetLastError(), "gemm_tn launch");
}

// ---------------------------------------------------------------------------------------------
// The fused training pass keeps gamma(x) / gamma(d) in B-operand slot order (swnerf_common.h sw_xs_col), so the
// weight-gradient GEMMs against them come out with slot-ordered columns: Cs[rows, nslots].  This moves every real
// slot to its reference column: W[o][col0 + sw_xs_col(slot0 + f)] = Cs[o][f]  (each column has exactly one slot;
// pad slots are dropped).
__global__ void __launch_bounds__(256) unslot_kernel(const float* Cs, int ld_s, int rows, int slot0, int nslots, int Lp, int Ld,
                                                     float* W, int ldw, int col0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nslots) return;
    const int o = idx / nslots, f = idx - o * nslots;
    const int col = sw_xs_col(slot0 + f, Lp, Ld);
    if (col >= 0) W[(size_t)o * ldw + col0 + col] = Cs[(size_t)o * ld_s + f];
}

extern "C" int swnerf_unslot_grad(const float* Cs, int ld_s, int rows, int slot0, int nslots, int L_pos, int L_dir,
                                  float* W, int ldw, int col0, void* stream) {
    if (!Cs || !W || rows < 1 || nslots < 1 || slot0 < 0 || slot0 + nslots > SW_XS_LD || ld_s < nslots)
        return sw_fail(SWNERF_E_ARG, "unslot_grad: bad arguments (rows=%d slot0=%d nslots=%d ld_s=%d)", rows, slot0, nslots, ld_s);
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "unslot_grad: embedder bands (%d,%d) exceed (10,4)", L_pos, L_dir);
    const int total = rows * nslots;
    hipLaunchKernelGGL(unslot_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Cs, ld_s, rows, slot0,
                       nslots, L_pos, L_dir, W, ldw, col0);
    return sw_check(hipGetLastError(), "unslot_grad launch");
}

// ---------------------------------------------------------------------------------------------
// The FIVE narrow weight-gradient products of the canonical net's fused training pass in ONE pass over the rows
// (swnerf_canon_narrow_grads).  As separate GEMMs they read 2.1 GB per 393 216-row chunk at ~3.2 TB/s with the matrix pipe idle
// (G and alpha_linear both read h7, G and the gamma(d) columns both read d pre_hv, pts_linears.0 and the gamma(d) columns
// both read the xs rows); together they are 64 accumulator tiles - exactly a 256 x 256 GEMM's - over 1.35 GB:
//   c0s [256, 64] += d pre_0^T . xs[:, :64]      pts_linears.0, gamma(x) slots            waves 8..11, 2 x 2 tiles each
//   cvs [128, 32] += d pre_hv^T . xs[:, 64:96]   views_linears.0, gamma(d) slots          wave 12, 4 x 1
//   G   [128,256] += d pre_hv^T . h7             (swnerf_feature_finish)                  waves 0..7, 2 x 2
//   a4w [4, 256]  += d raw^T . h7                alpha_linear = row 3                     waves 13, 14, 1 x 4
//   rgb4 [4, 128] += d raw^T . hv                rgb_linear = rows 0..2                   wave 15, 1 x 4
// and the column sums of d pre_0, d pre_hv and d raw (the biases).  16-row slabs of all six operands (55 KB), double buffered,
// filled by LDS-DMA; one barrier per slab; every wave issues 4 MFMAs per k-pair.
#define N5_SLAB 16
#define N5_A0 0                                   // d pre_0   [16][256]
#define N5_B1 (N5_SLAB * 256)                     // h7        [16][256]
#define N5_A1 (2 * N5_SLAB * 256)                 // d pre_hv  [16][128]
#define N5_B2 (N5_A1 + N5_SLAB * 128)             // hv        [16][128]
#define N5_B0 (N5_B2 + N5_SLAB * 128)             // xs        [16][96]
#define N5_A2 (N5_B0 + N5_SLAB * 96)              // d raw     [16][4]  (the DMA instruction writes 1 KiB: 256 floats reserved)
#define N5_BUF (N5_A2 + 256)
// (A three-deep ring for the two 1-KiB-per-row operands - 145 KB of LDS, two of their slabs in flight - measured the same
// 0.58 ms per 393 216 rows: the kernel is not bound by the latency of the one slab in flight but by the per-slab barrier and
// issue overhead of 16-row slabs, 5.4 us per slab against 3.7 us of MFMAs; 32-row slabs do not fit the LDS.)
struct Narrow5 {
    const float* grad; int ldg; const float* act; int lda; const float* xs; const float* d_out;
    int64_t M; int64_t rows_per_wg;
    float* c0s; float* cvs; float* G; float* a4w; float* rgb4; float* b_l0; float* b_hv; float* a4b; float* rgb4b;
};

// One 16-row slab of one wave's four tiles, with COMPILE-TIME operand pitches: every LDS read is base + immediate offset, the
// k-pairs are fully unrolled, and nothing but the MFMAs (and, on the waves that own bias entries, two adds) sits between the
// reads - round 3's loop carried 4 selects, 4 adds and 8 pointer increments per 4 MFMAs, and VALU instructions between MFMAs
// cost far more than their issue slots (profiles/r04/gemm_exp.md).  NA / NB: distinct A / B column blocks among the four tiles
// (2 x 2 block: tile k = A block k>>1 x B block k&1; 4 x 1: A block k; 1 x 4: B block k) - 4 or 5 LDS reads per k-pair, not 8.
// Rows past the slice are zero on the A side (the caller zeroes them once); lanes past a 4-column A operand compute rows of the
// tile that are never written.
template <int AP, int BP, int NA, int NB>
__device__ __forceinline__ void n5_slab(const float* lds, int ia, int ib, const int (&acol)[4], const int (&bcol)[4], f32x16 (&acc)[4],
                                        float& bs0, float& bs2, bool do_bias) {
    int pa[NA], pb[NB];
    float a[NA], b[NB];
#pragma unroll
    for (int x = 0; x < NA; ++x) { pa[x] = ia + acol[NA == 2 ? 2 * x : x]; a[x] = lds[pa[x]]; }
#pragma unroll
    for (int y = 0; y < NB; ++y) { pb[y] = ib + bcol[y]; b[y] = lds[pb[y]]; }
#pragma unroll
    for (int s = 0; s < N5_SLAB / 2; ++s) {
        float c[NA], d[NB];
#pragma unroll
        for (int x = 0; x < NA; ++x) c[x] = a[x];
#pragma unroll
        for (int y = 0; y < NB; ++y) d[y] = b[y];
        if (s + 1 < N5_SLAB / 2) {
#pragma unroll
            for (int x = 0; x < NA; ++x) a[x] = lds[pa[x] + 2 * (s + 1) * AP];
#pragma unroll
            for (int y = 0; y < NB; ++y) b[y] = lds[pb[y] + 2 * (s + 1) * BP];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (do_bias) { bs0 += c[0]; bs2 += c[NA == 2 ? 1 : 0]; }     // column sums of the A blocks of tiles 0 and 2 (wave-uniform branch)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(c[NA == 4 ? k : (NA == 2 ? k >> 1 : 0)], d[NB == 4 ? k : (NB == 2 ? k & 1 : 0)], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ void __launch_bounds__(1024) narrow5_kernel(Narrow5 P) {
    extern __shared__ __attribute__((aligned(16))) float n5_lds[];          // [2][N5_BUF]
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, hp = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * P.rows_per_wg;
    const int mlen = (int)(min(P.M, m0 + P.rows_per_wg) - m0);
    const int nslab = (mlen + N5_SLAB - 1) / N5_SLAB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)n5_lds);
    // DMA duties of wave w per slab: row w of d pre_0 and of h7; two 512-B rows of d pre_hv (w < 8) or hv (w >= 8); 1 KiB of
    // the slab's (contiguous) xs rows (w < 6); the d raw rows (w == 15).  Rows past the slice are clamped to its last row.
    // Slabs are issued in order, so the row pointers of a FULL slab advance by constants and its per-lane offsets are fixed
    // (all 16 waves run this right behind the barrier with the matrix pipe idle: 64-bit multiplies and a division per lane cost
    // there); only the slice's last, partial slab clamps its rows to the last one.
    const int kk = w & 7;
    const char* p_a0 = reinterpret_cast<const char*>(P.grad + (m0 + w) * P.ldg);
    const char* p_b1 = reinterpret_cast<const char*>(P.act + (m0 + w) * P.lda + 1792);
    const char* p_s = reinterpret_cast<const char*>((w < 8 ? P.grad + (m0 + 2 * kk) * P.ldg : P.act + (m0 + 2 * kk) * P.lda) + 2304);
    const char* p_x = reinterpret_cast<const char*>(P.xs + m0 * SW_XS_LD);
    const char* p_d = reinterpret_cast<const char*>(P.d_out + m0 * 4);
    const int64_t st_g = (int64_t)N5_SLAB * P.ldg * 4, st_a = (int64_t)N5_SLAB * P.lda * 4;
    const unsigned pitch_s = (unsigned)((w < 8 ? P.ldg : P.lda) * 4);
    const unsigned vo_s = (unsigned)(lane & 31) * 16u + (lane >> 5 ? pitch_s : 0u);
    const unsigned vo_x = (unsigned)(w * 1024 + lane * 16), vo_d = (unsigned)(min(lane, N5_SLAB - 1) * 16);
    auto issue = [&](int sl) {
        const int r0 = sl * N5_SLAB;
        const unsigned buf = lds0 + (unsigned)((sl & 1) * N5_BUF * 4);
        if (r0 + N5_SLAB <= mlen) {
            ws_dma(p_a0, (unsigned)lane * 16u, buf + (unsigned)((N5_A0 + w * 256) * 4));
            ws_dma(p_b1, (unsigned)lane * 16u, buf + (unsigned)((N5_B1 + w * 256) * 4));
            ws_dma(p_s, vo_s, buf + (unsigned)(((w < 8 ? N5_A1 : N5_B2) + kk * 256) * 4));
            if (w < 6) ws_dma(p_x, vo_x, buf + (unsigned)((N5_B0 + w * 256) * 4));
            if (w == 15) ws_dma(p_d, vo_d, buf + (unsigned)(N5_A2 * 4));
            p_a0 += st_g; p_b1 += st_a; p_s += (w < 8 ? st_g : st_a); p_x += N5_SLAB * SW_XS_LD * 4; p_d += N5_SLAB * 16;
            return;
        }
        auto rowc = [&](int r) { return min(r0 + r, mlen - 1); };           // slice-relative, clamped
        const int64_t rw = m0 + rowc(w);
        ws_dma(reinterpret_cast<const char*>(P.grad + rw * P.ldg), (unsigned)lane * 16u, buf + (unsigned)((N5_A0 + w * 256) * 4));
        ws_dma(reinterpret_cast<const char*>(P.act + rw * P.lda + 1792), (unsigned)lane * 16u, buf + (unsigned)((N5_B1 + w * 256) * 4));
        {
            const int k = w & 7, ra = rowc(2 * k), rb = rowc(2 * k + 1);
            const float* base = (w < 8 ? P.grad + (m0 + ra) * P.ldg : P.act + (m0 + ra) * P.lda) + 2304;
            const unsigned pitch = (unsigned)((w < 8 ? P.ldg : P.lda) * 4);
            const unsigned voff = (unsigned)(lane & 31) * 16u + (lane >> 5 ? (unsigned)(rb - ra) * pitch : 0u);
            ws_dma(reinterpret_cast<const char*>(base), voff, buf + (unsigned)(((w < 8 ? N5_A1 : N5_B2) + k * 256) * 4));
        }
        if (w < 6) {                                                         // xs rows are dense (384 B): flat KiB w of the slab
            const int flat = w * 1024 + lane * 16, r = flat / 384, off = flat - r * 384;
            const int ra = rowc(0);
            ws_dma(reinterpret_cast<const char*>(P.xs + (m0 + ra) * SW_XS_LD), (unsigned)((rowc(r) - ra) * 384 + off), buf + (unsigned)((N5_B0 + w * 256) * 4));
        }
        if (w == 15) {
            const int ra = rowc(0);
            ws_dma(reinterpret_cast<const char*>(P.d_out + (m0 + ra) * 4), (unsigned)((rowc(min(lane, N5_SLAB - 1)) - ra) * 16), buf + (unsigned)(N5_A2 * 4));
        }
    };
    // tile k of wave w multiplies columns acol[k].. of its A operand (LDS offset asrc) by columns bcol[k].. of its B operand - wave-uniform
    // scalars; the operand PITCHES are compile-time per role (n5_slab): the slab loop below dispatches on the role once per slab
    int asrc, bsrc, acol[4], bcol[4];
    if (w < 8) {                                             // G: A = d pre_hv tiles 2p, 2p+1; B = h7 tiles 2q, 2q+1
        asrc = N5_A1; bsrc = N5_B1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { acol[k] = 64 * (w & 1) + 32 * (k >> 1); bcol[k] = 64 * (w >> 1) + 32 * (k & 1); }
    } else if (w < 12) {                                     // pts_linears.0: A = d pre_0 tiles 2(w-8), +1; B = xs tiles 0, 1
        asrc = N5_A0; bsrc = N5_B0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { acol[k] = 64 * (w - 8) + 32 * (k >> 1); bcol[k] = 32 * (k & 1); }
    } else if (w == 12) {                                    // gamma(d) columns: A = d pre_hv tiles 0..3; B = xs tile 2
        asrc = N5_A1; bsrc = N5_B0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { acol[k] = 32 * k; bcol[k] = 64; }
    } else {                                                 // d raw (4 columns) x h7 tiles 4(w-13).. (w = 13, 14) or hv tiles 0..3 (w = 15)
        asrc = N5_A2; bsrc = w < 15 ? N5_B1 : N5_B2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { acol[k] = 0; bcol[k] = (w < 15 ? 128 * (w - 13) : 0) + 32 * k; }
    }
    // which waves own bias entries (column sums of their A blocks): d pre_hv -> waves 0, 1; d pre_0 -> 8..11; d raw -> 13 and 15
    const bool do_bias = (w < 2 && P.b_hv) || (w >= 8 && w < 12 && P.b_l0) || (w == 13 && P.a4b) || (w == 15 && P.rgb4b);
    f32x16 acc[4];
    float bs0 = 0.f, bs2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    // The WHOLE slab loop once per role (the role test in front of it, not inside): with the dispatch inside the loop hipcc keeps
    // the 64 accumulator registers alive across five code paths and spills 388 B per lane at this kernel's 128-register budget.
    // Every copy executes the same barriers, so the waves of a workgroup stay in step whichever copy they run.
    auto run = [&](auto ap_, auto bp_, auto na_, auto nb_, bool bias) {
        constexpr int AP = decltype(ap_)::value, BP = decltype(bp_)::value, NA = decltype(na_)::value, NB = decltype(nb_)::value;
        issue(0);
#pragma nounroll
        for (int sl = 0; sl < nslab; ++sl) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of slab sl has landed ...
#ifndef N5_EXP_NOBARRIER                                          // (-DN5_EXP_*: timing experiments, WRONG results - tools/probe_n5_exp.py)
            __syncthreads();                                      // ... everyone's has; and everyone is done with slab sl-1
#endif
            const int valid = mlen - sl * N5_SLAB;
            const int buf = (sl & 1) * N5_BUF;
            if (valid < N5_SLAB) {
                // the slice's last, partial slab: its rows past the slice hold clamped copies - zero them on the A side (d pre_0,
                // d pre_hv, d raw) once, instead of masking every operand of every k-pair
                for (int e = t; e < (N5_SLAB - valid) * 256; e += 1024) n5_lds[buf + N5_A0 + valid * 256 + e] = 0.f;
                for (int e = t; e < (N5_SLAB - valid) * 128; e += 1024) n5_lds[buf + N5_A1 + valid * 128 + e] = 0.f;
                if (t < (N5_SLAB - valid) * 4) n5_lds[buf + N5_A2 + valid * 4 + t] = 0.f;
                __syncthreads();
            }
#ifdef N5_EXP_NODMA
            if (sl + 1 < 2) issue(sl + 1);
#else
            if (sl + 1 < nslab) issue(sl + 1);
#endif
#ifndef N5_EXP_NOMFMA
            n5_slab<AP, BP, NA, NB>(n5_lds, buf + asrc + i + hp * AP, buf + bsrc + i + hp * BP, acol, bcol, acc, bs0, bs2, bias);
#endif
        }
    };
    using std::integral_constant;
    if (w < 8) run(integral_constant<int, 128>{}, integral_constant<int, 256>{}, integral_constant<int, 2>{}, integral_constant<int, 2>{}, do_bias);
    else if (w < 12) run(integral_constant<int, 256>{}, integral_constant<int, SW_XS_LD>{}, integral_constant<int, 2>{}, integral_constant<int, 2>{}, do_bias);
    else if (w == 12) run(integral_constant<int, 128>{}, integral_constant<int, SW_XS_LD>{}, integral_constant<int, 4>{}, integral_constant<int, 1>{}, false);
    else if (w < 15) run(integral_constant<int, 4>{}, integral_constant<int, 256>{}, integral_constant<int, 1>{}, integral_constant<int, 4>{}, do_bias);
    else run(integral_constant<int, 4>{}, integral_constant<int, 128>{}, integral_constant<int, 1>{}, integral_constant<int, 4>{}, do_bias);
    bs0 += __shfl_xor(bs0, 32, 64); bs2 += __shfl_xor(bs2, 32, 64);        // rows 2s and 2s+1 sit in the two lane halves
    float* C; int ldc, rlim = 256, cshift = 0;
    if (w < 8) { C = P.G; ldc = 256; }
    else if (w < 12) { C = P.c0s; ldc = 64; }
    else if (w == 12) { C = P.cvs; ldc = 32; cshift = 64; }
    else if (w < 15) { C = P.a4w; ldc = 256; rlim = 4; }
    else { C = P.rgb4; ldc = 128; rlim = 4; }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = acol[k] + sw_frow(r, hp);
            if (o < rlim) atomicAdd(C + (size_t)o * ldc + bcol[k] - cshift + i, acc[k][r]);
        }
    // column sums of the A operands: one wave per A tile writes them (tiles k = 0 and k = 2 of a 2 x 2 block hold different A tiles)
    if (hp == 0) {
        if (w < 2 && P.b_hv) { atomicAdd(P.b_hv + acol[0] + i, bs0); atomicAdd(P.b_hv + acol[2] + i, bs2); }
        if (w >= 8 && w < 12 && P.b_l0) { atomicAdd(P.b_l0 + acol[0] + i, bs0); atomicAdd(P.b_l0 + acol[2] + i, bs2); }
        if (w == 13 && P.a4b && i < 4) atomicAdd(P.a4b + i, bs0);
        if (w == 15 && P.rgb4b && i < 4) atomicAdd(P.rgb4b + i, bs0);
    }
}

extern "C" int swnerf_canon_narrow_grads(const float* grad, int ldg, const float* act, int lda, const float* xs, const float* d_out, int64_t M,
                                         float* c0s, float* cvs, float* G, float* a4w, float* rgb4, float* b_l0, float* b_hv, float* a4b,
                                         float* rgb4b, void* stream) {
    if (M == 0) return 0;
    if (!grad || !act || !xs || !d_out || !c0s || !cvs || !G || !a4w || !rgb4 || M < 0 || ldg < SW_ACT_LD || lda < SW_ACT_LD)
        return sw_fail(SWNERF_E_ARG, "canon_narrow_grads: NULL pointer, negative M or a leading dimension below %d", SW_ACT_LD);
    if (((uintptr_t)grad | (uintptr_t)act | (uintptr_t)xs | (uintptr_t)d_out) % 16 || ldg % 4 || lda % 4)
        return sw_fail(SWNERF_E_ARG, "canon_narrow_grads: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
    Narrow5 P;
    P.grad = grad; P.ldg = ldg; P.act = act; P.lda = lda; P.xs = xs; P.d_out = d_out; P.M = M;
    P.c0s = c0s; P.cvs = cvs; P.G = G; P.a4w = a4w; P.rgb4 = rgb4; P.b_l0 = b_l0; P.b_hv = b_hv; P.a4b = a4b; P.rgb4b = rgb4b;
    int64_t nwg = 256;
    int64_t rows = ((M + nwg - 1) / nwg + N5_SLAB - 1) / N5_SLAB * N5_SLAB;
    nwg = (M + rows - 1) / rows;
    P.rows_per_wg = rows;
    if (rows * (int64_t)(ldg > lda ? ldg : lda) * 4 >= (1LL << 31)) return sw_fail(SWNERF_E_UNSUPP, "canon_narrow_grads: row slice too large for 32-bit byte offsets");
    hipLaunchKernelGGL(narrow5_kernel, dim3((unsigned)nwg), dim3(1024), 2 * N5_BUF * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "canon_narrow_grads launch");
}

// ---------------------------------------------------------------------------------------------
// The same idea for ANY set of narrow products over the same rows, table driven (round 4): the deformation net's
// (`_time.0` against gamma(x) and gamma(t), `_time_out`) and the no-view net's (pts_linears.0 against gamma(x), output_linear)
// used to be two or three skinny GEMM launches per chunk, each re-reading d pre_0 or h7.  A plan names up to NP_MAX_OPS operand
// windows (pointer, leading dimension, staged width), gives every wave ONE A operand and ONE B operand with four (A column,
// B column) tile offsets - the single code path of narrow5_kernel, whose role tables are data here - and deals the slab's
// 1-KiB DMA pieces out over the 16 waves.  NP_SLAB-row slabs, double buffered by LDS-DMA, one barrier per slab.
// Measured (profiles/r04/narrow_plan.md): the deformation net's set 183 us per 196 608-row chunk in ONE launch on the main stream
// against ~205 us for its three skinny GEMMs; the no-view net's set 304 us against 254 us - two padded products on 8 of 16 waves
// are matrix-pipe bound there (16 + 8 tiles of 64-cycle MFMAs per row pair on four SIMDs) - so that net keeps its GEMMs.
#ifndef NP_SLAB
#define NP_SLAB 16                // rows per slab (32 fit the LDS for the plans below but measured 5-10 % slower: profiles/r04/narrow_plan.md)
#endif
#define NP_MAX_OPS 4
#define NP_MAX_JOBS 5
struct NpOp { const float* ptr; int ld; int width; int lds_off; int kib; };      // width: floats staged per row (multiple of 4); kib: 1-KiB pieces per slab image
struct NpWave {
    short a_op, b_op;              // operand indices; a_op < 0: this wave only helps with the DMA
    short shape;                   // which compile-time tile shape its four tiles have (narrow_plan_kernel: 1..4)
    short rlim, cshift, bias_lim;  // output rows < rlim; C column = B column - cshift; bias entries < bias_lim per block
    short acol[4], bcol[4];
    float* C; int ldc; int bias_mask;   // bit k: the column sums of this wave's A block k go to bias[acol[k] + i]
    float* bias;
    unsigned char job_op[NP_MAX_JOBS]; unsigned char job_kib[NP_MAX_JOBS];      // DMA duty: piece job_kib of operand job_op (255: none)
};
struct NpPlan { int64_t M, rows_per_wg; int buf_floats; int a_ops; NpOp op[NP_MAX_OPS]; NpWave wave[16]; };   // a_ops: bit o = operand o is an A side

__global__ void __launch_bounds__(1024) narrow_plan_kernel(NpPlan P) {
    extern __shared__ __attribute__((aligned(16))) float np_lds[];          // [2][buf_floats]
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, hp = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const NpWave& R = P.wave[w];
    const int64_t m0 = (int64_t)blockIdx.x * P.rows_per_wg;
    const int mlen = (int)(min(P.M, m0 + P.rows_per_wg) - m0);
    const int nslab = (mlen + NP_SLAB - 1) / NP_SLAB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)np_lds);
    // DMA duties: piece q of operand o = bytes [1024 q, 1024 q + 1024) of the slab's dense [NP_SLAB][width] image; lane's 16 bytes
    // sit in image row `jrow`, at byte `jcb` of it.  Per slab the wave-uniform base advances by NP_SLAB rows; only the slice's last,
    // partial slab clamps its rows.  Lanes past the image (its size is not always a multiple of 1 KiB) re-read its first bytes.
    const char* jbase[NP_MAX_JOBS];
    unsigned jvoff[NP_MAX_JOBS], jlds[NP_MAX_JOBS], jrow[NP_MAX_JOBS], jcb[NP_MAX_JOBS], jpitch[NP_MAX_JOBS];
    int64_t jstep[NP_MAX_JOBS];
    bool jon[NP_MAX_JOBS];
#pragma unroll
    for (int j = 0; j < NP_MAX_JOBS; ++j) {
        jon[j] = R.job_op[j] != 255;
        const NpOp& O = P.op[jon[j] ? R.job_op[j] : 0];
        const unsigned rowb = (unsigned)O.width * 4u, b = (unsigned)R.job_kib[j] * 1024u + (unsigned)lane * 16u;
        const bool in = b < rowb * NP_SLAB;
        jrow[j] = in ? b / rowb : 0u;
        jcb[j] = in ? b - jrow[j] * rowb : 0u;
        jpitch[j] = (unsigned)O.ld * 4u;
        jvoff[j] = jrow[j] * jpitch[j] + jcb[j];
        jbase[j] = reinterpret_cast<const char*>(O.ptr + m0 * O.ld);
        jstep[j] = (int64_t)NP_SLAB * O.ld * 4;
        jlds[j] = (unsigned)(O.lds_off * 4) + (unsigned)R.job_kib[j] * 1024u;
    }
    auto issue = [&](int sl) {
        const unsigned buf = lds0 + (unsigned)((sl & 1) * P.buf_floats * 4);
        const int valid = mlen - sl * NP_SLAB;
#pragma unroll
        for (int j = 0; j < NP_MAX_JOBS; ++j) {
            if (!jon[j]) continue;                                   // wave-uniform
            const unsigned vo = valid >= NP_SLAB ? jvoff[j] : min(jrow[j], (unsigned)(valid - 1)) * jpitch[j] + jcb[j];
            ws_dma(jbase[j], vo, buf + jlds[j]);
            jbase[j] += jstep[j];
        }
    };
    const bool active = R.a_op >= 0;
    const NpOp& OA = P.op[active ? R.a_op : 0];
    const NpOp& OB = P.op[active ? R.b_op : 0];
    const int asrc = OA.lds_off, bsrc = OB.lds_off;
    int acol[4], bcol[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { acol[k] = R.acol[k]; bcol[k] = R.bcol[k]; }
    f32x16 acc[4];
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const bool do_bias = R.bias != nullptr;
    // the whole slab loop once per tile shape (operand pitches and block pattern are compile-time inside: n5_slab), the shape test
    // in front of the loop - see narrow5_kernel.  shape 0: idle (DMA only)
    auto run = [&](auto ap_, auto bp_, auto na_, auto nb_) {
        constexpr int AP = decltype(ap_)::value, BP = decltype(bp_)::value, NA = decltype(na_)::value, NB = decltype(nb_)::value;
        issue(0);
#pragma nounroll
        for (int sl = 0; sl < nslab; ++sl) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of slab sl has landed ...
            __syncthreads();                                      // ... everyone's has; and everyone is done with slab sl-1
            const int valid = mlen - sl * NP_SLAB;
            const int buf = (sl & 1) * P.buf_floats;
            if (valid < NP_SLAB) {                                // the slice's last, partial slab: zero the rows past it on the A side(s)
                for (int o = 0; o < NP_MAX_OPS; ++o) {
                    if (!((P.a_ops >> o) & 1)) continue;
                    const int wd = P.op[o].width;
                    for (int e = t; e < (NP_SLAB - valid) * wd; e += 1024) np_lds[buf + P.op[o].lds_off + valid * wd + e] = 0.f;
                }
                __syncthreads();
            }
            if (sl + 1 < nslab) issue(sl + 1);
            if constexpr (NA > 0) n5_slab<AP, BP, NA, NB>(np_lds, buf + asrc + i + hp * AP, buf + bsrc + i + hp * BP, acol, bcol, acc, bs[0], bs[2], do_bias);
        }
    };
    using std::integral_constant;
    switch (active ? R.shape : 0) {
        case 1: run(integral_constant<int, 256>{}, integral_constant<int, SW_XS_LD>{}, integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
        case 2: run(integral_constant<int, 256>{}, integral_constant<int, SW_XS_LD>{}, integral_constant<int, 4>{}, integral_constant<int, 1>{}); break;
        case 3: run(integral_constant<int, 4>{}, integral_constant<int, 256>{}, integral_constant<int, 1>{}, integral_constant<int, 4>{}); break;
        case 4: run(integral_constant<int, 8>{}, integral_constant<int, 256>{}, integral_constant<int, 1>{}, integral_constant<int, 4>{}); break;
        default: run(integral_constant<int, 4>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{}); break;
    }
    if (!active) return;
    bs[0] += __shfl_xor(bs[0], 32, 64); bs[2] += __shfl_xor(bs[2], 32, 64);  // rows 2s and 2s+1 sit in the two lane halves
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = acol[k] + sw_frow(r, hp);
            if (o < R.rlim) atomicAdd(R.C + (size_t)o * R.ldc + bcol[k] - R.cshift + i, acc[k][r]);
        }
    if (hp == 0 && R.bias && i < R.bias_lim) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((R.bias_mask >> k) & 1) atomicAdd(R.bias + acol[k] + i, bs[k]);
    }
}

// host side: lay the operands out in the slab buffer, deal the DMA pieces out over the waves, launch
struct NpBuilder {
    NpPlan P; int n_ops, cursor;
    NpBuilder(int64_t M) : n_ops(0), cursor(0) {
        P.M = M; P.a_ops = 0;
        for (int w = 0; w < 16; ++w) {
            NpWave& R = P.wave[w];
            R.a_op = -1; R.b_op = 0; R.shape = 0; R.rlim = 0; R.cshift = 0; R.bias_lim = 32; R.C = nullptr; R.ldc = 0; R.bias_mask = 0; R.bias = nullptr;
            for (int k = 0; k < 4; ++k) { R.acol[k] = 0; R.bcol[k] = 0; }
            for (int j = 0; j < NP_MAX_JOBS; ++j) { R.job_op[j] = 255; R.job_kib[j] = 0; }
        }
        for (int o = 0; o < NP_MAX_OPS; ++o) { P.op[o].ptr = nullptr; P.op[o].ld = 0; P.op[o].width = 4; P.op[o].lds_off = 0; P.op[o].kib = 0; }
    }
    int op(const float* ptr, int ld, int width) {
        NpOp& O = P.op[n_ops];
        O.ptr = ptr; O.ld = ld; O.width = width; O.lds_off = cursor;
        O.kib = (NP_SLAB * width * 4 + 1023) / 1024;
        cursor += O.kib * 256;                               // whole KiB pieces: a DMA instruction always writes 1 KiB
        return n_ops++;
    }
    // shape: 1 = 256-wide A x xs, 2 x 2 block | 2 = 256-wide A x xs, 4 x 1 | 3 = 4-column A x 256-wide B, 1 x 4 | 4 = 8-column A x 256-wide B, 1 x 4
    NpWave& wave(int w, int shape, int a_op, int b_op, float* C, int ldc, int rlim) {
        NpWave& R = P.wave[w];
        R.shape = (short)shape; R.a_op = (short)a_op; R.b_op = (short)b_op; R.C = C; R.ldc = ldc; R.rlim = (short)rlim;
        P.a_ops |= 1 << a_op;
        return R;
    }
    int launch(const char* what, void* stream) {
        int w = 0, slot[16] = {0};
        for (int o = 0; o < n_ops; ++o)
            for (int q = 0; q < P.op[o].kib; ++q) {
                if (slot[w] == NP_MAX_JOBS) return sw_fail(SWNERF_E_ARG, "%s: too many DMA pieces per slab for the plan kernel", what);
                P.wave[w].job_op[slot[w]] = (unsigned char)o; P.wave[w].job_kib[slot[w]] = (unsigned char)q;
                ++slot[w];
                w = (w + 1) & 15;
            }
        P.buf_floats = cursor + 2 * 256;                     // + the rows a last prefetch may touch
        int64_t nwg = 256;
        int64_t rows = ((P.M + nwg - 1) / nwg + NP_SLAB - 1) / NP_SLAB * NP_SLAB;
        nwg = (P.M + rows - 1) / rows;
        P.rows_per_wg = rows;
        int maxld = 0;
        for (int o = 0; o < n_ops; ++o) maxld = P.op[o].ld > maxld ? P.op[o].ld : maxld;
        if (rows * (int64_t)maxld * 4 >= (1LL << 31)) return sw_fail(SWNERF_E_UNSUPP, "%s: row slice too large for 32-bit byte offsets", what);
        hipLaunchKernelGGL(narrow_plan_kernel, dim3((unsigned)nwg), dim3(1024), 2 * (size_t)P.buf_floats * sizeof(float), (hipStream_t)stream, P);
        return sw_check(hipGetLastError(), what);
    }
};

static bool np_aligned(const void* p) { return ((uintptr_t)p % 16) == 0; }

// Deformation net of DirectTemporalNeRF (model.py:128-136), fused D-NeRF training pass: over the M rows of a chunk
//   c0s [256, 64] += d pre_0^T . xs_d[:, :64]     `_time.0`, gamma(x) slots        b_l0 [256] += column sums of d pre_0
//   cts [256, 32] += d pre_0^T . xs_d[:, 64:96]   `_time.0`, gamma(t) slots
//   w4  [4, 256]  += g_dx^T . h7                  `_time_out` = rows 0..2           b4 [4] += column sums of g_dx
// grad_d / act_d: [M, ld >= 2432] (d pre_0 at column 0, h7 at 1792), xs_d [M, 96], g_dx [M, 4] (4th column zero).
extern "C" int swnerf_deform_narrow_grads(const float* grad_d, int ldg, const float* act_d, int lda, const float* xs_d, const float* g_dx, int64_t M,
                                          float* c0s, float* cts, float* w4, float* b_l0, float* b4, void* stream) {
    if (M == 0) return 0;
    if (!grad_d || !act_d || !xs_d || !g_dx || !c0s || !cts || !w4 || M < 0 || ldg < SW_ACT_LD || lda < SW_ACT_LD)
        return sw_fail(SWNERF_E_ARG, "deform_narrow_grads: NULL pointer, negative M or a leading dimension below %d", SW_ACT_LD);
    if (!np_aligned(grad_d) || !np_aligned(act_d) || !np_aligned(xs_d) || !np_aligned(g_dx) || ldg % 4 || lda % 4)
        return sw_fail(SWNERF_E_ARG, "deform_narrow_grads: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
    NpBuilder B(M);
    const int o_g = B.op(grad_d, ldg, 256), o_x = B.op(xs_d, SW_XS_LD, SW_XS_LD), o_h = B.op(act_d + 1792, lda, 256), o_d = B.op(g_dx, 4, 4);
    for (int w = 0; w < 4; ++w) {                                                // gamma(x) slots: A tiles 2w, 2w+1 x xs tiles 0, 1
        NpWave& R = B.wave(w, 1, o_g, o_x, c0s, 64, 256);
        for (int k = 0; k < 4; ++k) { R.acol[k] = (short)(64 * w + 32 * (k >> 1)); R.bcol[k] = (short)(32 * (k & 1)); }
        R.bias = b_l0; R.bias_mask = 0x5;
    }
    for (int w = 4; w < 6; ++w) {                                                // gamma(t) slots: A tiles 4(w-4)..+3 x xs tile 2
        NpWave& R = B.wave(w, 2, o_g, o_x, cts, 32, 256);
        for (int k = 0; k < 4; ++k) { R.acol[k] = (short)(128 * (w - 4) + 32 * k); R.bcol[k] = 64; }
        R.cshift = 64;
    }
    for (int w = 6; w < 8; ++w) {                                                // _time_out: d dx (4 columns) x h7 tiles 4(w-6)..+3
        NpWave& R = B.wave(w, 3, o_d, o_h, w4, 256, 4);
        for (int k = 0; k < 4; ++k) { R.acol[k] = 0; R.bcol[k] = (short)(128 * (w - 6) + 32 * k); }
        if (w == 6) { R.bias = b4; R.bias_mask = 0x1; R.bias_lim = 4; }
    }
    return B.launch("deform_narrow_grads launch", stream);
}

// The net without view directions (model.py:59-60), fused training pass:
//   c0s [256, 64] += d pre_0^T . xs[:, :64]       pts_linears.0, gamma(x) slots      b_l0 [256] += column sums of d pre_0
//   w8  [8, 256]  += d_raw8^T . h7                output_linear = rows 0..out_ch-1   b8 [8] += column sums of d_raw8
// grad / act: [M, ld >= 2048 + ...] as above, xs [M, 96], d_raw8 [M, 8] (columns >= out_ch zero).
extern "C" int swnerf_noview_narrow_grads(const float* grad, int ldg, const float* act, int lda, const float* xs, const float* d_raw8, int64_t M,
                                          float* c0s, float* w8, float* b_l0, float* b8, void* stream) {
    if (M == 0) return 0;
    if (!grad || !act || !xs || !d_raw8 || !c0s || !w8 || M < 0 || ldg < SW_ACT_LD || lda < SW_ACT_LD)
        return sw_fail(SWNERF_E_ARG, "noview_narrow_grads: NULL pointer, negative M or a leading dimension below %d", SW_ACT_LD);
    if (!np_aligned(grad) || !np_aligned(act) || !np_aligned(xs) || !np_aligned(d_raw8) || ldg % 4 || lda % 4)
        return sw_fail(SWNERF_E_ARG, "noview_narrow_grads: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
    NpBuilder B(M);
    const int o_g = B.op(grad, ldg, 256), o_x = B.op(xs, SW_XS_LD, SW_XS_LD), o_h = B.op(act + 1792, lda, 256), o_d = B.op(d_raw8, 8, 8);
    for (int w = 0; w < 4; ++w) {
        NpWave& R = B.wave(w, 1, o_g, o_x, c0s, 64, 256);
        for (int k = 0; k < 4; ++k) { R.acol[k] = (short)(64 * w + 32 * (k >> 1)); R.bcol[k] = (short)(32 * (k & 1)); }
        R.bias = b_l0; R.bias_mask = 0x5;
    }
    for (int w = 4; w < 6; ++w) {                                                // output_linear: d raw (8 columns) x h7 tiles 4(w-4)..+3
        NpWave& R = B.wave(w, 4, o_d, o_h, w8, 256, 8);
        for (int k = 0; k < 4; ++k) { R.acol[k] = 0; R.bcol[k] = (short)(128 * (w - 4) + 32 * k); }
        if (w == 4) { R.bias = b8; R.bias_mask = 0x1; R.bias_lim = 8; }
    }
    return B.launch("noview_narrow_grads launch", stream);
}

// ---------------------------------------------------------------------------------------------
// feature_linear has no activation (model.py:50-51), so the fused training pass never stores `feature` or d feature and never
// runs feature_linear's 256 x 256 weight-gradient GEMM: with G = sum_rows d pre_hv (x) h7 [128, 256] (one narrow GEMM) and
// db_hv = sum_rows d pre_hv,
//   d views_linears.0.weight[:, :256] += G . W_f^T + db_hv (x) b_f      (feature = W_f h7 + b_f)
//   d feature_linear.weight          += Wv_f^T . G                      (d feature = Wv_f^T d pre_hv; Wv_f = views_linears.0.weight[:, :256])
//   d feature_linear.bias            += Wv_f^T . db_hv
// and alpha_linear's gradient is row 3 of the 4-row form (A = d raw [rows, 4]).  25 MFLOP once per backward pass: one small launch.
__global__ void __launch_bounds__(256) feature_finish_kernel(const float* G, const float* db_hv, const float* Wv, int ldwv, const float* W_f,
                                                             const float* b_f, const float* a4w, const float* a4b, float* dWv, int ld_dwv,
                                                             float* dW_f, float* db_f, float* dW_alpha, float* db_alpha) {
    __shared__ float sh[256];
    const int t = threadIdx.x, b = blockIdx.x;
    if (b < 128) {                                           // row u = b of d views_linears.0.weight[:, :256]; thread = output column o
        sh[t] = G[b * 256 + t];
        __syncthreads();
        const float* w = W_f + (size_t)t * 256;
        float acc = 0.f;
        for (int i = 0; i < 256; i += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + i);
            acc = fmaf(sh[i], w4[0], acc); acc = fmaf(sh[i + 1], w4[1], acc); acc = fmaf(sh[i + 2], w4[2], acc); acc = fmaf(sh[i + 3], w4[3], acc);
        }
        dWv[(size_t)b * ld_dwv + t] += acc + db_hv[b] * b_f[t];
    } else if (b < 384) {                                    // row o = b - 128 of d feature_linear.weight; thread = column i
        const int o = b - 128;
        if (t < 128) sh[t] = Wv[(size_t)t * ldwv + o];
        __syncthreads();
        float acc = 0.f;
        for (int u = 0; u < 128; ++u) acc = fmaf(sh[u], G[u * 256 + t], acc);
        dW_f[(size_t)o * 256 + t] += acc;
        if (t == 0) {
            float bb = 0.f;
            for (int u = 0; u < 128; ++u) bb = fmaf(sh[u], db_hv[u], bb);
            db_f[o] += bb;
        }
    } else {
        dW_alpha[t] += a4w[3 * 256 + t];
        if (t == 0) db_alpha[0] += a4b[3];
    }
}

extern "C" int swnerf_feature_finish(const float* G, const float* db_hv, const float* Wv, int ldwv, const float* W_f, const float* b_f,
                                     const float* a4w, const float* a4b, float* dWv, int ld_dwv, float* dW_f, float* db_f,
                                     float* dW_alpha, float* db_alpha, void* stream) {
    if (!G || !db_hv || !Wv || !W_f || !b_f || !a4w || !a4b || !dWv || !dW_f || !db_f || !dW_alpha || !db_alpha || ldwv < 256 || ld_dwv < 256)
        return sw_fail(SWNERF_E_ARG, "feature_finish: NULL pointer or a leading dimension below 256");
    if (((uintptr_t)W_f) % 16) return sw_fail(SWNERF_E_ARG, "feature_finish: feature_linear.weight must be 16-byte aligned");
    hipLaunchKernelGGL(feature_finish_kernel, dim3(385), dim3(256), 0, (hipStream_t)stream, G, db_hv, Wv, ldwv, W_f, b_f, a4w, a4b, dWv, ld_dwv,
                       dW_f, db_f, dW_alpha, db_alpha);
    return sw_check(hipGetLastError(), "feature_finish launch");
}

// xs_d of the fused D-NeRF training pass carries gamma(t) in its third k-tile: slot f (0..31) -> sw_time_col
__global__ void __launch_bounds__(256) unslot_time_kernel(const float* Cs, int ld_s, int rows, int nslots, int Lt, float* W, int ldw, int col0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nslots) return;
    const int o = idx / nslots, f = idx - o * nslots;
    const int g = (f >> 3) & 3, h = (f >> 2) & 1, e = f & 3;
    const int col = sw_time_col(4 * g + e, h, Lt);
    if (col >= 0) W[(size_t)o * ldw + col0 + col] = Cs[(size_t)o * ld_s + f];
}

extern "C" int swnerf_unslot_grad_time(const float* Cs, int ld_s, int rows, int nslots, int L_time, float* W, int ldw, int col0, void* stream) {
    if (!Cs || !W || rows < 1 || nslots < 1 || nslots > 32 || ld_s < nslots || L_time < 0 || L_time > 10)
        return sw_fail(SWNERF_E_ARG, "unslot_grad_time: bad arguments (rows=%d nslots=%d ld_s=%d L_time=%d)", rows, nslots, ld_s, L_time);
    const int total = rows * nslots;
    hipLaunchKernelGGL(unslot_time_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Cs, ld_s, rows, nslots, L_time, W, ldw, col0);
    return sw_check(hipGetLastError(), "unslot_grad_time launch");
}
