// mlp_core16.h - the 8x256 NeRF MLP on 16-row tiles: v_mfma_f32_16x16x4_f32, half the registers of the 32-row design
// (mlp_core.h), so that TWO wavefronts fit a SIMD and the VALU / LDS / scalar work of one runs under the MFMAs of the
// other - which one wave cannot do for itself (profiles/r01/mfma_shadow_microbench.md, profiles/r02/
// two_waves_microbench.md).  Same ideas as mlp_core.h: the transposed problem H_out^T[256 x 16] = W[256 x K].H_in^T,
// weights = A operand, activations = B operand, accumulator registers ARE the next layer's B operand, weights stream
// through a per-wave LDS-DMA ring in 1-KiB steps (4 MFMAs each).
//
// v_mfma_f32_16x16x4_f32:  A lane (i = l&15, kq = l>>4): A[i][kq];  B lane (j = l&15, g = l>>4): B[k = g][j];
// C/D lane (j, g) register r: row 4g + r, column j.  So after a layer lane (j, g) holds, per 16-feature tile n,
// the features 16n + 4g + r (r = 0..3) of row j; as the B operand of MFMA q of k-tile kt it supplies k-slot g with
// register q, i.e. feature 16kt + 4g + q - a bijection per q, and the packed weight for that slot is
// W[..][16kt + 4kq + q]: four consecutive q are 16 CONTIGUOUS bytes of the torch weight row.
// A step interleaves two output tiles (n, n+1) x two q so that consecutive MFMAs never share an accumulator (the
// 16x16x4 form has 40 cycles of dependent latency on a 32-cycle issue): element e of the step's float4 belongs to
// tile 2np + (e&1), q = 2half + (e>>1).
// Biases and head weights sit in LDS in NATURAL feature order: lane group g reads floats [16n + 4g, +4).
//
// Reference arithmetic: model.py:39-62 (vallina_NeRF.forward), :273-296 (NeRFOriginal).
#pragma once
#include <hip/hip_runtime.h>
#include "swnerf_common.h"
#include "mlp_core.h"          // WStream, ws_*, relu1, f32x4 (sw-nerf_amd/csrc, on the include path)

// ---- slot maps of the 16-row design (host + device; the pack kernel uses the same functions) ---------------------
// position encoding: 4 k-tiles x 4 q = 16 slots s per lane group g.  Groups 0/1 evaluate sin/cos of the arguments
// a = s (bands 0..4), groups 2/3 of a = 15 + s (bands 5..9), a = 3k + c; slot 15 carries x itself (g = 0,1,2) / pad.
SW_HD int sw16_pos_col(int s /*0..15*/, int g, int L) {
    if (s < 15) { const int a = s + 15 * (g >> 1), k = a / 3, c = a % 3; return k < L ? 3 + 6 * k + 3 * (g & 1) + c : -1; }
    return g < 3 ? g : -1;
}
// view direction: 2 k-tiles x 4 q = 8 slots.  Groups 0/1: sin/cos of a = s (8 arguments); groups 2/3: a = 8 + s for
// s < 4; then slot 4: d0 (g=2) / d1 (g=3), slot 5: d2 (g=2).
SW_HD int sw16_dir_col(int s /*0..7*/, int g, int L) {
    if (g < 2) { const int k = s / 3, c = s % 3; return k < L ? 3 + 6 * k + 3 * g + c : -1; }
    if (s < 4) { const int a = 8 + s, k = a / 3, c = a % 3; return k < L ? 3 + 6 * k + 3 * (g & 1) + c : -1; }
    if (s == 4) return g == 2 ? 0 : 1;
    if (s == 5) return g == 2 ? 2 : -1;
    return -1;
}

#define SW16_BIAS_FLOATS (8 * 256 + 256 + 16 + 256 + 128 + 384)     // L0..L7 | alpha w | head biases | FEAT | VIEWS | rgb w
#define SW16_W_FLOATS ((SW_CANON_STEPS + SW_TAIL) * SW_STEP_FLOATS)
#define SW16_FLOATS (SW16_W_FLOATS + SW16_BIAS_FLOATS)

#if defined(__HIPCC__)
enum { S16_ACC = 0, S16_BIAS = 1 };

// out[n] (+)= sum_kt Wtile(n,kt) . kin[kt]   NT (even) output tiles of 16 features, KT input tiles of 16.
// ws.bias: LDS, this lane group's 4 floats of the CURRENT segment's first bias tile (natural order: + 16 n per tile).
template <int S, int NS, int NT, int KT, int INIT>
__device__ __forceinline__ void seg16_steps(f32x4 (&out)[NT], const f32x4 (&kin)[KT], WStream& ws, f32x4& b0, f32x4& b1) {
    if constexpr (S < NS) {
        constexpr int np = S / (KT * 2), kt = (S / 2) % KT, half = S % 2;
        constexpr int slot = S % SW_RING, nslot = (S + 1) % SW_RING;
        constexpr bool pair_first = (S % (KT * 2)) == 0, pair_last = ((S + 1) % (KT * 2)) == 0;
        if constexpr (pair_first && INIT == S16_BIAS) { out[2 * np] = b0; out[2 * np + 1] = b1; }
        ws_wait<SW_RING - 2>();
        const f32x4 a_next = ws_read(ws, nslot);
        if constexpr (pair_last && np + 1 < NT / 2 && INIT == S16_BIAS) {
            b0 = *reinterpret_cast<const f32x4*>(ws.bias + 16 * (2 * np + 2));
            b1 = *reinterpret_cast<const f32x4*>(ws.bias + 16 * (2 * np + 3));
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 a = ws.a_cur;
        out[2 * np] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], kin[kt][2 * half], out[2 * np], 0, 0, 0);
        out[2 * np + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], kin[kt][2 * half], out[2 * np + 1], 0, 0, 0);
        out[2 * np] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], kin[kt][2 * half + 1], out[2 * np], 0, 0, 0);
        out[2 * np + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], kin[kt][2 * half + 1], out[2 * np + 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ws_dma(ws.base + (S + SW_RING) * 1024, ws.voff, ws.lds_addr + slot * 1024);
        ws.a_cur = a_next;
        seg16_steps<S + 1, NS, NT, KT, INIT>(out, kin, ws, b0, b1);
    }
}

template <int NT, int KT, int INIT>
__device__ __forceinline__ void seg16(f32x4 (&out)[NT], const f32x4 (&kin)[KT], WStream& ws) {
    constexpr int NS = (NT / 2) * KT * 2;
    static_assert(NT % 2 == 0 && NS % SW_RING == 0, "segment must keep the ring phase");
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if constexpr (INIT == S16_BIAS) {
        b0 = *reinterpret_cast<const f32x4*>(ws.bias);
        b1 = *reinterpret_cast<const f32x4*>(ws.bias + 16);
    }
    seg16_steps<0, NS, NT, KT, INIT>(out, kin, ws, b0, b1);
    if (INIT == S16_BIAS) ws.bias += 16 * NT;
    ws.base += NS * 1024;
}

// res[o] = sum_f W[o][f] x[f] over NT*16 features; W rows in natural order at ws.bias (+ 4g); every lane gets the sum.
template <int NOUT, int NT>
__device__ __forceinline__ void head16(const f32x4 (&x)[NT], WStream& ws, float (&res)[NOUT]) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(ws.bias + (o * NT + n) * 16);
            acc = fmaf(w[0], x[n][0], acc); acc = fmaf(w[1], x[n][1], acc);
            acc = fmaf(w[2], x[n][2], acc); acc = fmaf(w[3], x[n][3], acc);
        }
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        res[o] = acc;
        asm volatile("" : "+v"(res[o]));
    }
    ws.bias += NOUT * NT * 16;
}

__device__ __forceinline__ void pe16_pos(float x0, float x1, float x2, int g, f32x4 (&e)[4]) {
    const float sc = (g >> 1) ? 32.f : 1.f;             // groups 2/3 take the bands 5..9: 2^(k+5) = 32 * 2^k, exact
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float v;
        if (s < 15) {
            const int k = s / 3, c = s % 3;
            const float xc = (c == 0) ? x0 : ((c == 1) ? x1 : x2);
            v = sw_sin_or_cos(xc * (float)(1 << k) * sc, g & 1);
        } else {
            v = (g == 0) ? x0 : ((g == 1) ? x1 : ((g == 2) ? x2 : 0.f));
        }
        e[s >> 2][s & 3] = v;
    }
}

__device__ __forceinline__ void pe16_dir(float d0, float d1, float d2, int g, f32x4 (&e)[2]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // groups 0/1: argument a = s; groups 2/3: a = 8 + s (s < 4).  a = 3k + c
        const int a_lo = s, a_hi = 8 + s;
        const int klo = a_lo / 3, clo = a_lo % 3, khi = a_hi / 3, chi = a_hi % 3;
        const float dlo = (clo == 0) ? d0 : ((clo == 1) ? d1 : d2), dhi = (chi == 0) ? d0 : ((chi == 1) ? d1 : d2);
        float v;
        if (s < 4) {
            const float arg = (g >> 1) ? dhi * (float)(1 << khi) : dlo * (float)(1 << klo);
            v = sw_sin_or_cos(arg, g & 1);
        } else {
            const float sv = sw_sin_or_cos(dlo * (float)(1 << klo), g & 1);
            const float raw = (s == 4) ? ((g == 2) ? d0 : d1) : ((s == 5 && g == 2) ? d2 : 0.f);
            v = (g >> 1) ? raw : sv;
        }
        e[s >> 2][s & 3] = v;
    }
}

// 8 layers of width 256 with the skip at layer 5; on return `in` = relu(layer 7), sigma on every lane.
// g4 = 4 * lane group (ws.bias carries that offset; the head-bias floats are read without it).
__device__ __forceinline__ void trunk16(const f32x4 (&emb)[4], f32x4 (&in)[16], f32x4 (&out)[16], float& sigma, WStream& ws, int g4) {
#pragma nounroll
    for (int l = 0; l < 8; ++l) {
        if (l == 0) {
            seg16<16, 4, S16_BIAS>(out, emb, ws);
        } else {
            seg16<16, 16, S16_BIAS>(out, in, ws);
            if (l == 5) seg16<16, 4, S16_ACC>(out, emb, ws);                 // skip: cat[input_pts, h] (model.py:45-46)
        }
#pragma unroll
        for (int n = 0; n < 16; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) in[n][r] = relu1(out[n][r]);
    }
    float s1[1];
    head16<1, 16>(in, ws, s1);
    sigma = s1[0] + (ws.bias - g4)[0];                                        // head biases: [b_alpha, b_r, b_g, b_b, 0 x 12]
    ws.bias += 16;
}

// feature_linear (no activation) -> views_linears[0] + relu -> rgb_linear.  hb: the head-bias floats in LDS.
__device__ __forceinline__ void tail16(const f32x4 (&in)[16], f32x4 (&out)[16], const f32x4 (&demb)[2], float (&rgb)[3],
                                       const float* hb, WStream& ws) {
    seg16<16, 16, S16_BIAS>(out, in, ws);
    f32x4 k18[18];
#pragma unroll
    for (int n = 0; n < 16; ++n) k18[n] = out[n];
    k18[16] = demb[0]; k18[17] = demb[1];
    f32x4 hv[8];
    seg16<8, 18, S16_BIAS>(hv, k18, ws);
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[n][r] = relu1(hv[n][r]);
    head16<3, 8>(hv, ws, rgb);
    rgb[0] += hb[1]; rgb[1] += hb[2]; rgb[2] += hb[3];
}
#endif
