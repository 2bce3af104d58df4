// train_kernels.hip - the training path of the MLP: the forward that saves activations
// and ReLU bit masks, the dX chains of the canonical and the deformation net.  Same building blocks as the render
// kernels (mlp_core.h) but with a 16-deep weight ring: these kernels also issue stores, `vmcnt` retires in order, and
// a store's write acknowledge takes longer than 6 ring steps often enough to cost 8 % (measured: forward 7.57 ->
// 6.94 ms, dX chain 6.64 -> 6.32 ms at 786 k rows).  The render kernels keep 8: their LDS also holds 28 KB of
// resampling scratch, and depth made no difference there.
#ifndef SW_RING
#define SW_RING 16
#endif
#include "mlp_kernels.h"
#include "render_pass.h"

// ------------------------------------------------------------------------------------------
// Backward of the MLP w.r.t. its activations (the "dX chain"), one wave per 32 rows, mirroring the
// forward: transposed weight streams, the same register-resident accumulator->operand hand-over.  The ReLU
// derivative comes from the forward's bit masks, fetched one layer ahead by LDS-DMA into a 1-KiB slot of the
// wave (nothing the compiler sees is ever pending, like the weight ring); d(pre-activation) of every layer goes
// to grad[M, SW_ACT_LD] (same column map as act) for the weight-gradient GEMMs as side stores of the segment
// that consumes it (mlp_core.h SideStore).   model.py:39-62 reversed.
struct MaskRing {
    const char* base;        // wave-uniform: this tile's SW_MASK_TILE_FLOATS of bit masks
    unsigned voff, lds_addr; // lane * 16; LDS byte address of the slot
    const float* slot;       // the slot + lane * 4 floats
};
__device__ __forceinline__ void mask_start(MaskRing& mr, const float* bits, int64_t tile, float* lds_slot, int lane) {
    mr.base = reinterpret_cast<const char*>(bits + tile * SW_MASK_TILE_FLOATS);
    mr.voff = (unsigned)lane * 16u;
    mr.lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_slot);
    mr.slot = lds_slot + lane * 4;
}
// fetch layer l's mask; the previous contents must have been read (mask_take ends with lgkmcnt(0))
__device__ __forceinline__ void mask_fetch(const MaskRing& mr, int l) { ws_dma(mr.base + l * 1024, mr.voff, mr.lds_addr); }
// the mask fetched last; legal once a counted wait behind >= SW_RING later DMAs has passed (any full segment)
__device__ __forceinline__ f32x4 mask_take(const MaskRing& mr) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(mr.slot);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return v;
}

struct DxDev {
    const float* w0; const float* b0;      // backward stream; its 8 "bias" tiles = alpha_linear.weight
    const float* bits; const float* d_out; // [ceil(M/32), SW_MASK_TILE_FLOATS], [M,4]
    int64_t M; float* grad;
    const float* pts; float* d_pts; int Lp; // INGRAD: the embedded positions [M,3], their gradient [M,3]
};

// INGRAD: also d gamma(x) = pts_linears.5.weight[:, :Cpos]^T . d pre_5 + pts_linears.0.weight^T . d pre_0 in the
// slots pe_pos() fills (pack_kernels.hip rowmap), then d x = J_gamma(x)^T . d gamma(x) on the VALU (embedder.py:33-42).
template <bool INGRAD>
__global__ void __launch_bounds__(256, 1) mlp_backward_dx_kernel(DxDev P) {
    extern __shared__ __attribute__((aligned(16))) float lds_bias[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds_ring = lds_bias + SW_LDS_BIAS_FLOATS + wv * SW_LDS_RING_FLOATS;
    float* lds_emb = lds_ring + SW_RING * SW_STEP_FLOATS;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    bias_to_lds(lds_bias, P.b0, SW_BWD_BIAS_TILES * SW_BIAS_TILE_FLOATS);
    if (tile * 32 >= P.M) return;
    const int64_t row = tile * 32 + j;
    const bool live = row < P.M;
    const int64_t rr = live ? row : P.M - 1;
    const f32x4 dr = *reinterpret_cast<const f32x4*>(P.d_out + rr * 4);     // d rgb(3), d sigma
    float* grad_row = P.grad + rr * SW_ACT_LD + 4 * h;
    const f32x4 nomask = {0.f, 0.f, 0.f, 0.f};
    MaskRing mr;
    mask_start(mr, P.bits, tile, lds_emb + 2 * 16 * 64, lane);
    mask_fetch(mr, 8);                                                   // views hidden; older than every weight step
    WStream ws;
    ws_start(ws, P.w0, lds_bias, lds_ring, lane);
    // d hv = rgb_linear.weight^T . d rgb, masked by hv > 0           (4 tiles <- 1 k-tile holding 3 channels)
    f32x16 k1[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) k1[0][r] = 0.f;
    k1[0][0] = h ? 0.f : dr[0]; k1[0][1] = h ? 0.f : dr[1]; k1[0][2] = h ? 0.f : dr[2];
    f32x16 dhv[4];
    seg_mfma<4, 1, SEG_ZERO>(dhv, k1, ws);
    mask_apply<4>(mask_take(mr), dhv);
    mask_fetch(mr, 7);
    // d h7 = W_vf^T . d pre_hv + alpha_linear.weight * d sigma  (W_vf = views_linears.0.weight[:, :256] . feature_linear.weight:
    // no activation on feature_linear, so the two transposed layers are one, swnerf_common.h SW_BWD_STEPS), masked by h7 > 0 below
    f32x16 in[8], out[8];
    seg_mfma<8, 4, SEG_BIAS_SCALED, 4>(out, dhv, ws, dr[3], SideStore{grad_row + SW_ACT_HV, nullptr, nomask});
#pragma nounroll
    for (int l = 7; l >= 1; --l) {
        // out = d h_l;  d pre_l = out . [h_l > 0];  d h_{l-1} = W_l[:, -256:]^T . d pre_l
#pragma unroll
        for (int n = 0; n < 8; ++n) in[n] = out[n];
        mask_apply<8>(mask_take(mr), in);
        mask_fetch(mr, l - 1);
        if (INGRAD && l == 5) {                                          // the skip input cat[gamma(x), h4]: its gamma(x) part
            f32x16 ge[2];
            seg_mfma<2, 8, SEG_ZERO>(ge, in, ws);
            emb_park(lds_emb, lane, ge);
        }
        seg_mfma<8, 8, SEG_ZERO, 8>(out, in, ws, 1.f, SideStore{grad_row + 256 * l, nullptr, nomask});
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) in[n] = out[n];
    mask_apply<8>(mask_take(mr), in);                                    // d pre_0
    if (!INGRAD) {
        tiles_store<8>(grad_row, in);
    } else {
        f32x16 ge[2];
        emb_fetch(lds_emb, lane, ge);
        seg_mfma<2, 8, SEG_ACC, 8>(ge, in, ws, 1.f, SideStore{grad_row, nullptr, nomask});
        // slot a of lane half h holds sin (h=0) / cos (h=1) of 2^k x_c, a = 3k + c < 30; slots 30, 31 hold x itself
        const float x0 = P.pts[rr * 3], x1 = P.pts[rr * 3 + 1], x2 = P.pts[rr * 3 + 2];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int a = 0; a < 30; ++a) {
            const int k = a / 3, c = a % 3;
            const float f = (float)(1 << k);
            const float xc = (c == 0) ? x0 : ((c == 1) ? x1 : x2);
            float d = sw_sin_or_cos(xc * f, 1 - h) * f;                  // d sin = cos, d cos = -sin
            d = h ? -d : d;
            const float t = (k < P.Lp) ? ge[a >> 4][a & 15] * d : 0.f;
            if (c == 0) g0 += t; else if (c == 1) g1 += t; else g2 += t;
        }
        if (h == 0) { g0 += ge[1][14]; g1 += ge[1][15]; } else { g2 += ge[1][14]; }
        g0 += __shfl_xor(g0, 32, 64); g1 += __shfl_xor(g1, 32, 64); g2 += __shfl_xor(g2, 32, 64);
        if (live && h == 0) { P.d_pts[row * 3] = g0; P.d_pts[row * 3 + 1] = g1; P.d_pts[row * 3 + 2] = g2; }
    }
}

// ------------------------------------------------------------------------------------------
// DirectTemporalNeRF training: the deformation net alone, saving its activations (model.py:128-136).
struct DeformDev {
    const float* x; const float* t_emb; int64_t M; int C, Lp, Ct;
    const float* w0; const float* b0; int nbias;
    float* dx; float* act; float* bits;
};

__global__ void __launch_bounds__(256, 1) deform_forward_train_kernel(DeformDev P) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    extern __shared__ __attribute__((aligned(16))) float lds_bias[];
    float* lds_ring = lds_bias + SW_LDS_BIAS_FLOATS + wv * SW_LDS_RING_FLOATS;
    float* lds_emb = lds_ring + SW_RING * SW_STEP_FLOATS;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    bias_to_lds(lds_bias, P.b0, P.nbias);
    if (tile * 32 >= P.M) return;
    const int64_t row = tile * 32 + j;
    const bool live = row < P.M;
    const int64_t rr = live ? row : P.M - 1;
    const float* xr = P.x + rr * P.C;
    f32x16 emb[2], in[8], out[8];
    float head[3];
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        const int col = sw_pos_col(a, h, P.Lp);
        emb[a >> 4][a & 15] = (col >= 0) ? xr[col] : 0.f;
    }
    const float ft = P.t_emb[rr * P.Ct];                                  // column 0 of gamma(t) is t
    WStream ws;
    ws_start(ws, P.w0 + SW_STEPS_DIR * SW_STEP_FLOATS, lds_bias + SW_DIR_BIAS_TILES * SW_BIAS_TILE_FLOATS, lds_ring, lane);   // (behind the per-ray DIR prefix)
    trunk_pass<true, true>(emb, lds_emb, ft, true, h, in, out, head, ws, P.act + rr * SW_ACT_LD + 4 * h,
                           P.bits + tile * SW_MASK_TILE_FLOATS + lane * 4, true, nullptr);
    if (live && h == 0) { P.dx[row * 3] = head[0]; P.dx[row * 3 + 1] = head[1]; P.dx[row * 3 + 2] = head[2]; }
}

// dX chain of the deformation net: d h7 = _time_out.weight^T . d dx, then _time.7 .. _time.1 (model.py:128-136 reversed)
struct DeformBwdDev {
    const float* w0; const float* b0;      // SWNERF_BWD_DEFORM stream; its 24 "bias" tiles = _time_out.weight rows
    const float* bits; const float* d_dx;  // [ceil(M/32), SW_MASK_TILE_FLOATS], [M,3]
    int64_t M; float* grad;
};

__global__ void __launch_bounds__(256, 1) deform_backward_dx_kernel(DeformBwdDev P) {
    extern __shared__ __attribute__((aligned(16))) float lds_bias[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds_ring = lds_bias + SW_LDS_BIAS_FLOATS + wv * SW_LDS_RING_FLOATS;
    float* lds_emb = lds_ring + SW_RING * SW_STEP_FLOATS;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    bias_to_lds(lds_bias, P.b0, SW_DBWD_BIAS_TILES * SW_BIAS_TILE_FLOATS);
    if (tile * 32 >= P.M) return;
    const int64_t row = tile * 32 + j;
    const bool live = row < P.M;
    const int64_t rr = live ? row : P.M - 1;
    const float d0 = P.d_dx[rr * 3], d1 = P.d_dx[rr * 3 + 1], d2 = P.d_dx[rr * 3 + 2];
    float* grad_row = P.grad + rr * SW_ACT_LD + 4 * h;
    const f32x4 nomask = {0.f, 0.f, 0.f, 0.f};
    MaskRing mr;
    mask_start(mr, P.bits, tile, lds_emb + 2 * 16 * 64, lane);
    mask_fetch(mr, 7);
    WStream ws;
    ws_start(ws, P.w0, lds_bias, lds_ring, lane);
    f32x16 in[8], out[8];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws.bias + (0 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(ws.bias + (1 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(ws.bias + (2 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[n][4 * g + e] = w0[e] * d0 + w1[e] * d1 + w2[e] * d2;
        }
#pragma nounroll
    for (int l = 7; l >= 1; --l) {
#pragma unroll
        for (int n = 0; n < 8; ++n) in[n] = out[n];
        mask_apply<8>(mask_take(mr), in);
        mask_fetch(mr, l - 1);
        seg_mfma<8, 8, SEG_ZERO, 8>(out, in, ws, 1.f, SideStore{grad_row + 256 * l, nullptr, nomask});
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) in[n] = out[n];
    mask_apply<8>(mask_take(mr), in);                                    // d pre_0 (x and t are data: no further gradient)
    tiles_store<8>(grad_row, in);
}

// ------------------------------------------------------------------------------------------
// Backward of the fused render pass (static net): ONE wavefront owns ONE ray, like the forward.
//   1. compositing backward (autograd of ray.py:155-198) from d rgb_map / d disp_map / d acc_map: forward recompute of
//      T and w from the saved raw and depths, suffix sums of G.w in double, d raw of every sample -> the wave's LDS
//      slice (and, padded to whole tiles, to HBM: the rgb_linear / alpha_linear weight-gradient GEMMs read it);
//   2. per 32-sample tile the dX chain of mlp_backward_dx_kernel, seeded from LDS: d(pre-activation) of every layer
//      -> grad[rows, SW_ACT_LD] as side stores.
// Rows are the forward's padded rows: (ray * ntiles + tile) * 32 + j; samples past S get zero gradients.
#define PB_SMAX 256
#define PB_WAVE_FLOATS (6 * PB_SMAX)                 // T[S], w[S], d_raw[S][4]
struct PassBwdDev {
    const float* w0; const float* b0;                // backward stream; bias tiles: alpha_linear.weight (8) [, _time_out.weight rows (24)]
    const float* bits;                               // [n_rays * ntiles, SW_MASK_TILE_FLOATS]
    const float* raw; const float* z; const float* ray_batch; int cols; const float* noise;
    int64_t n_rays; int S; int white;
    const float* g_rgb; const float* g_disp; const float* g_acc;
    const float* g_raw;                              // upstream gradient of the returned raw [N,S,4] (retraw) or NULL: added to d raw
    float* grad; float* d_raw;                       // [rows, SW_ACT_LD], [rows, 4]
    // D-NeRF: the deformation net's masks, the saved dx = position_delta [N,S,3], its upstream gradient (or NULL),
    // the position-encoding band count, and the outputs grad_d [rows, SW_ACT_LD], g_dx [rows, 4] (= d dx, 4th column 0)
    const float* bits_d; const float* dx; const float* g_pd; int Lp;
    float* grad_d; float* g_dx;
    // net without view directions (MODE 2): channels of output_linear (4 or 5); raw / g_raw are [N,S,out_ch], d_raw is [rows, 8]
    int out_ch;
};

// MODE: 0 static net with view directions | 1 DirectTemporalNeRF | 2 static net WITHOUT view directions (the stream is
// pts_linears.7..1 transposed; output_linear.weight rides as out_ch x 8 bias-style tiles: d h7 = sum_c w_c . d raw_c on the VALU)
template <int MODE>
__global__ void __launch_bounds__(256, 1) render_pass_backward_kernel(PassBwdDev P) {
    constexpr bool DNERF = MODE == 1, NOVIEW = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    constexpr int BIASF = (NOVIEW ? SW_NVBWD_BIAS_TILES : (DNERF ? SW_BWD_BIAS_TILES + SW_DBWD_BIAS_TILES : SW_BWD_BIAS_TILES)) * SW_BIAS_TILE_FLOATS;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
    const int oc = NOVIEW ? P.out_ch : 4;
    bias_to_lds(lds_all, P.b0, NOVIEW ? oc * 8 * SW_BIAS_TILE_FLOATS : BIASF);
    if (ray >= P.n_rays) return;
    float* lds_ring = lds_all + BIASF + wv * SW_LDS_RING_FLOATS;
    float* lds_emb = lds_ring + SW_RING * SW_STEP_FLOATS;
    float* T_ = lds_all + BIASF + 4 * SW_LDS_RING_FLOATS + wv * PB_WAVE_FLOATS;
    float* W_ = T_ + PB_SMAX;
    float* dR = W_ + PB_SMAX;
    const int S = P.S;
    const int ntiles = (S + 31) >> 5;
    const float* rb = P.ray_batch + ray * P.cols;
    const float rox = rb[0], roy = rb[1], roz = rb[2], ddx = rb[3], ddy = rb[4], ddz = rb[5];
    const float dnorm = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
    const float* zv = P.z + ray * S;
    const float* raw = P.raw + ray * S * oc;
    WStream ws;
    ws_start(ws, P.w0, lds_all, lds_ring, lane);      // the weight ring fills while the compositing backward runs

    // ---- 1. compositing backward (the arithmetic of raw2outputs_bwd_kernel, backward_kernels.hip)
    const float gr = P.g_rgb ? P.g_rgb[ray * 3] : 0.f, gg = P.g_rgb ? P.g_rgb[ray * 3 + 1] : 0.f, gb = P.g_rgb ? P.g_rgb[ray * 3 + 2] : 0.f;
    double Tc = 1.0;
    float pa = 0.f, pd = 0.f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        const float z = zv[sc];
        float dist = (s + 1 < S) ? (zv[s + 1] - z) : 1e10f;
        dist *= dnorm;
        float sg = raw[sc * oc + 3];
        if (P.noise) sg += P.noise[ray * S + sc];
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
    float gA = P.g_acc ? P.g_acc[ray] : 0.f, gD = 0.f;
    if (P.white) gA -= (gr + gg + gb);
    if (P.g_disp) {
        const float q = pd / pa;                       // disp = 1/max(1e-10, q); no gradient on the clamped / NaN branch
        if (q > 1e-10f) { const float gq = -P.g_disp[ray] / (q * q); gD += gq / pa; gA -= gq * pd / (pa * pa); }
    }
    wave_lds_sync();
    double carry = 0.0;
    const int nch = (S + 63) / 64;
    for (int ch = nch - 1; ch >= 0; --ch) {
        const int s = ch * 64 + lane;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        f32x4 r4;
        if (NOVIEW) { r4[0] = raw[sc * oc]; r4[1] = raw[sc * oc + 1]; r4[2] = raw[sc * oc + 2]; r4[3] = raw[sc * oc + 3]; }   // rows of 5 floats are not 16-byte aligned
        else r4 = *reinterpret_cast<const f32x4*>(raw + sc * 4);
        const float z = zv[sc];
        const float c0 = 1.f / (1.f + expf(-r4[0])), c1 = 1.f / (1.f + expf(-r4[1])), c2 = 1.f / (1.f + expf(-r4[2]));
        const float w = live ? W_[sc] : 0.f, T = live ? T_[sc] : 0.f;
        const float G = gr * c0 + gg * c1 + gb * c2 + gA + gD * z;
        double v = live ? (double)G * (double)w : 0.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double dn = __shfl_down(v, o, 64); if (lane + o < 64) v += dn; }
        double after = __shfl_down(v, 1, 64);
        if (lane == 63) after = 0.0;
        const double R = carry + after;
        carry += __shfl(v, 0, 64);
        float dist = (s + 1 < S) ? (zv[s + 1] - z) : 1e10f;
        dist *= dnorm;
        float sg = r4[3];
        if (P.noise) sg += P.noise[ray * S + sc];
        const float e = expf(-fmaxf(sg, 0.f) * dist);
        const float p = 1.f - (1.f - e) + 1e-10f;
        const float dLda = G * T - (float)(R / (double)p);
        const float dsig = (sg > 0.f) ? dLda * dist * e : 0.f;
        if (live) {
            f32x4 o4 = {w * gr * c0 * (1.f - c0), w * gg * c1 * (1.f - c1), w * gb * c2 * (1.f - c2), dsig};
            *reinterpret_cast<f32x4*>(dR + 4 * s) = o4;
        }
    }
    wave_lds_sync();

    // ---- 2. the dX chain(s), tile by tile: mlp_backward_dx_kernel<DNERF> [+ deform_backward_dx_kernel]
    const f32x4 nomask = {0.f, 0.f, 0.f, 0.f};
#pragma nounroll
    for (int tile = 0; tile < ntiles; ++tile) {
        const int s = tile * 32 + j;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        const int64_t tix = ray * ntiles + tile;
        const int64_t prow = tix * 32 + j;
        f32x4 dr = *reinterpret_cast<const f32x4*>(dR + 4 * sc);
        float dr5 = 0.f;                                                     // NOVIEW, out_ch 5: the fifth channel gets a gradient from g_raw alone
        if (P.g_raw) {
            if (NOVIEW) {
                const float* gq = P.g_raw + (ray * S + sc) * oc;
                dr[0] += gq[0]; dr[1] += gq[1]; dr[2] += gq[2]; dr[3] += gq[3];
                if (oc == 5) dr5 = gq[4];
            } else {
                dr += *reinterpret_cast<const f32x4*>(P.g_raw + (ray * S + sc) * 4);
            }
        }
        if (!live) { dr = nomask; dr5 = 0.f; }
        if (h == 0) {
            if (NOVIEW) {                                                    // [rows, 8]: the A operand of output_linear's weight-gradient GEMM
                const f32x4 hi4 = {dr5, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(P.d_raw + prow * 8) = dr;
                *reinterpret_cast<f32x4*>(P.d_raw + prow * 8 + 4) = hi4;
            } else {
                *reinterpret_cast<f32x4*>(P.d_raw + prow * 4) = dr;
            }
        }
        float* grad_row = P.grad + prow * SW_ACT_LD + 4 * h;
        MaskRing mr;
        mask_start(mr, P.bits, tix, lds_emb + 2 * 16 * 64, lane);
        f32x16 in[8], out[8];
        if constexpr (NOVIEW) {
            mask_fetch(mr, 7);
            // d h7 = output_linear.weight^T . d raw: out_ch weight rows as bias-style tiles (tile n of channel c: [h][r])
            const float* wt = lds_all + h * 16;
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + (0 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wt + (1 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
                    const f32x4 w2 = *reinterpret_cast<const f32x4*>(wt + (2 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
                    const f32x4 w3 = *reinterpret_cast<const f32x4*>(wt + (3 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[n][4 * g + e] = w0[e] * dr[0] + w1[e] * dr[1] + w2[e] * dr[2] + w3[e] * dr[3];
                }
            if (oc == 5) {                                                   // wave-uniform
#pragma unroll
                for (int n = 0; n < 8; ++n)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + (4 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) out[n][4 * g + e] += w4[e] * dr5;
                    }
            }
            ws_wait<0>();            // h7's mask has landed (no segment lies between its fetch and its use here: once per tile, ~1 us)
        } else {
        mask_fetch(mr, 8);
        f32x16 k1[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) k1[0][r] = 0.f;
        k1[0][0] = h ? 0.f : dr[0]; k1[0][1] = h ? 0.f : dr[1]; k1[0][2] = h ? 0.f : dr[2];
        f32x16 dhv[4];
        seg_mfma<4, 1, SEG_ZERO>(dhv, k1, ws);
        mask_apply<4>(mask_take(mr), dhv);
        mask_fetch(mr, 7);
        // d h7 = W_vf^T . d pre_hv + alpha_linear.weight * d sigma: feature_linear folded into the view layer (SW_BWD_STEPS)
        seg_mfma<8, 4, SEG_BIAS_SCALED, 4>(out, dhv, ws, dr[3], SideStore{grad_row + SW_ACT_HV, nullptr, nomask});
        }
#pragma nounroll
        for (int l = 7; l >= 1; --l) {
#pragma unroll
            for (int n = 0; n < 8; ++n) in[n] = out[n];
            mask_apply<8>(mask_take(mr), in);
            mask_fetch(mr, l - 1);
            if (DNERF && l == 5) {                                           // the skip input cat[gamma(x+dx), h4]: its gamma part
                f32x16 ge[2];
                seg_mfma<2, 8, SEG_ZERO>(ge, in, ws);
                emb_park(lds_emb, lane, ge);
            }
            seg_mfma<8, 8, SEG_ZERO, 8>(out, in, ws, 1.f, SideStore{grad_row + 256 * l, nullptr, nomask});
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) in[n] = out[n];
        mask_apply<8>(mask_take(mr), in);                                    // d pre_0
        if (!DNERF) {
            tiles_store<8>(grad_row, in);
        } else {
            // d gamma(x+dx) -> d(x+dx) through the sin/cos Jacobian (mlp_backward_dx_kernel<true>), + the upstream gradient of
            // position_delta (the TV loss's operand, run_dnerf.py:700-716) = d dx: the seed of the deformation net's chain
            mask_start(mr, P.bits_d, tix, lds_emb + 2 * 16 * 64, lane);
            mask_fetch(mr, 7);                                               // h7 of the deformation net; taken after the next segment
            f32x16 ge[2];
            emb_fetch(lds_emb, lane, ge);
            seg_mfma<2, 8, SEG_ACC, 8>(ge, in, ws, 1.f, SideStore{grad_row, nullptr, nomask});
            const float z = zv[sc];
            const float* dxs = P.dx + (ray * S + sc) * 3;
            const float x0 = rox + ddx * z + dxs[0], x1 = roy + ddy * z + dxs[1], x2 = roz + ddz * z + dxs[2];   // x + dx, as the forward formed it
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
            for (int a = 0; a < 30; ++a) {
                const int k = a / 3, c = a % 3;
                const float f = (float)(1 << k);
                const float xc = (c == 0) ? x0 : ((c == 1) ? x1 : x2);
                float d = sw_sin_or_cos(xc * f, 1 - h) * f;                  // d sin = cos, d cos = -sin
                d = h ? -d : d;
                const float t = (k < P.Lp) ? ge[a >> 4][a & 15] * d : 0.f;
                if (c == 0) g0 += t; else if (c == 1) g1 += t; else g2 += t;
            }
            if (h == 0) { g0 += ge[1][14]; g1 += ge[1][15]; } else { g2 += ge[1][14]; }
            g0 += __shfl_xor(g0, 32, 64); g1 += __shfl_xor(g1, 32, 64); g2 += __shfl_xor(g2, 32, 64);
            if (P.g_pd) { const float* gp = P.g_pd + (ray * S + sc) * 3; g0 += gp[0]; g1 += gp[1]; g2 += gp[2]; }
            if (!live) { g0 = 0.f; g1 = 0.f; g2 = 0.f; }
            if (h == 0) { f32x4 g4 = {g0, g1, g2, 0.f}; *reinterpret_cast<f32x4*>(P.g_dx + prow * 4) = g4; }
            // d h7 = _time_out.weight^T . d dx  (24 bias-style tiles right behind alpha_linear's 8), then _time.7 .. _time.1
            float* gd_row = P.grad_d + prow * SW_ACT_LD + 4 * h;
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws.bias + (0 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(ws.bias + (1 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
                    const f32x4 w2 = *reinterpret_cast<const f32x4*>(ws.bias + (2 * 8 + n) * SW_BIAS_TILE_FLOATS + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[n][4 * g + e] = w0[e] * g0 + w1[e] * g1 + w2[e] * g2;
                }
#pragma nounroll
            for (int l = 7; l >= 1; --l) {
#pragma unroll
                for (int n = 0; n < 8; ++n) in[n] = out[n];
                mask_apply<8>(mask_take(mr), in);
                mask_fetch(mr, l - 1);
                seg_mfma<8, 8, SEG_ZERO, 8>(out, in, ws, 1.f, SideStore{gd_row + 256 * l, nullptr, nomask});
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) in[n] = out[n];
            mask_apply<8>(mask_take(mr), in);                                // d pre_0 of the deformation net (x, t are data)
            tiles_store<8>(gd_row, in);
        }
        ws_rewind(ws, P.w0, lds_all, lane);
    }
}

extern "C" size_t swnerf_packed_bwd_floats(void) { return (size_t)SW_BWD_FLOATS; }
extern "C" size_t swnerf_act_floats_per_row(void) { return (size_t)SW_ACT_LD; }
extern "C" size_t swnerf_mask_floats(int64_t M) { return M <= 0 ? 0 : (size_t)((M + 31) / 32) * SW_MASK_TILE_FLOATS; }

extern "C" int swnerf_mlp_backward_dx(const float* packed_bwd, const float* bits, const float* d_out, int64_t M,
                                      float* grad, void* stream) {
    if (M == 0 && packed_bwd) return 0;
    if (!packed_bwd || !bits || !d_out || !grad || M < 0) return sw_fail(SWNERF_E_ARG, "mlp_backward_dx: NULL pointer or negative M");
    DxDev P;
    P.w0 = packed_bwd; P.b0 = packed_bwd + SW_BWD_W_FLOATS; P.bits = bits; P.d_out = d_out; P.M = M; P.grad = grad;
    P.pts = nullptr; P.d_pts = nullptr; P.Lp = 0;
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    hipLaunchKernelGGL(mlp_backward_dx_kernel<false>, grid, block, SW_LDS_FIXED_FLOATS * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "mlp_backward_dx launch");
}

extern "C" int swnerf_mlp_backward_dx_pts(const float* packed_bwd, const float* bits, const float* d_out, const float* pts,
                                          int64_t M, int L_pos, float* grad, float* d_pts, void* stream) {
    if (M == 0 && packed_bwd) return 0;
    if (!packed_bwd || !bits || !d_out || !pts || !grad || !d_pts || M < 0)
        return sw_fail(SWNERF_E_ARG, "mlp_backward_dx_pts: NULL pointer or negative M");
    if (L_pos < 0 || L_pos > 10) return sw_fail(SWNERF_E_UNSUPP, "mlp_backward_dx_pts: %d position bands exceed 10", L_pos);
    DxDev P;
    P.w0 = packed_bwd; P.b0 = packed_bwd + SW_BWD_IG_W_FLOATS; P.bits = bits; P.d_out = d_out; P.M = M; P.grad = grad;
    P.pts = pts; P.d_pts = d_pts; P.Lp = L_pos;
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    hipLaunchKernelGGL(mlp_backward_dx_kernel<true>, grid, block, SW_LDS_FIXED_FLOATS * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "mlp_backward_dx_pts launch");
}

extern "C" int swnerf_deform_backward_dx(const float* packed_bwd, const float* bits_d, const float* d_dx, int64_t M,
                                         float* grad_d, void* stream) {
    if (M == 0 && packed_bwd) return 0;
    if (!packed_bwd || !bits_d || !d_dx || !grad_d || M < 0) return sw_fail(SWNERF_E_ARG, "deform_backward_dx: NULL pointer or negative M");
    DeformBwdDev P;
    P.w0 = packed_bwd; P.b0 = packed_bwd + SW_DBWD_W_FLOATS; P.bits = bits_d; P.d_dx = d_dx; P.M = M; P.grad = grad_d;
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    hipLaunchKernelGGL(deform_backward_dx_kernel, grid, block, SW_LDS_FIXED_FLOATS * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "deform_backward_dx launch");
}

extern "C" int swnerf_mlp_forward_train(const float* packed, const float* x, int64_t M, int L_pos, int L_dir,
                                        float* out, float* act, float* bits, void* stream) {
    if (M == 0 && packed) return 0;
    if (!packed || !x || !out || !act || !bits || M < 0) return sw_fail(SWNERF_E_ARG, "mlp_forward_train: NULL pointer or negative M");
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "mlp_forward_train: embedder bands (%d,%d) exceed (10,4)", L_pos, L_dir);
    MlpDev P;
    P.x = x; P.M = M; P.Lp = L_pos; P.Ld = L_dir; P.Lt = 0;
    P.Cpos = 3 * (1 + 2 * L_pos); P.C = P.Cpos + 3 * (1 + 2 * L_dir);
    P.t_emb = nullptr; P.Ct = 1; P.out = out; P.dx = nullptr; P.act = act; P.bits = bits;
    int rc = stream_ptrs(SWNERF_NET_CANON, packed, 0, &P.w0, &P.b0, &P.nbias, &P.two_pass);
    if (rc) return rc;
    P.wvl = views_loop_ptr(SWNERF_NET_CANON, packed);
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    hipLaunchKernelGGL((mlp_forward_kernel<false, true>), grid, block, SW_LDS_FIXED_FLOATS * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "mlp_forward_train launch");
}

extern "C" int swnerf_deform_forward_train(const float* packed, const float* x, const float* t_emb, int64_t M,
                                           int L_pos, int L_dir, int L_time, float* dx, float* act_d, float* bits_d, void* stream) {
    if (M == 0 && packed) return 0;
    if (!packed || !x || !t_emb || !dx || !act_d || !bits_d || M < 0) return sw_fail(SWNERF_E_ARG, "deform_forward_train: NULL pointer or negative M");
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4 || L_time < 0 || L_time > 10)
        return sw_fail(SWNERF_E_UNSUPP, "deform_forward_train: embedder bands (%d,%d,%d) exceed (10,4,10)", L_pos, L_dir, L_time);
    DeformDev P;
    P.x = x; P.t_emb = t_emb; P.M = M; P.Lp = L_pos; P.C = 3 * (1 + 2 * L_pos) + 3 * (1 + 2 * L_dir); P.Ct = 1 + 2 * L_time;
    P.dx = dx; P.act = act_d; P.bits = bits_d;
    int two = 0;
    int rc = stream_ptrs(SWNERF_NET_DNERF, packed, 1, &P.w0, &P.b0, &P.nbias, &two);
    if (rc) return rc;
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    hipLaunchKernelGGL(deform_forward_train_kernel, grid, block, SW_LDS_FIXED_FLOATS * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "deform_forward_train launch");
}

// ---- the fused training pass ("backward for the fused path") ---------------------
extern "C" int64_t swnerf_train_rows(int64_t n_rays, int n_samples) { return n_rays * (int64_t)((n_samples + 31) / 32) * 32; }
extern "C" int swnerf_xs_floats_per_row(void) { return SW_XS_LD; }

extern "C" int swnerf_render_pass_train(const swnerf_pass_args* args, float* act, float* bits, float* xs, void* stream) {
    if (!args) return sw_fail(SWNERF_E_ARG, "render_pass_train: NULL args");
    const swnerf_pass_args& a = *args;
    if (a.n_rays == 0 && a.packed) return 0;
    if (!a.packed || !a.ray_batch || !act || !bits || !xs) return sw_fail(SWNERF_E_ARG, "render_pass_train: NULL pointer");
    const bool noview = a.kind == SWNERF_NET_NOVIEW;
    if (noview ? a.cols != 8 : (a.kind != SWNERF_NET_CANON || (a.cols != 11 && a.cols != 12)))
        return sw_fail(SWNERF_E_UNSUPP, "render_pass_train: the static net (SWNERF_NET_CANON, 11- or 12-column ray batch) or the one without view directions (SWNERF_NET_NOVIEW, 8 columns)");
    if (a.n_rays < 0 || a.n_samples < 2 || a.n_samples > PB_SMAX) return sw_fail(SWNERF_E_UNSUPP, "render_pass_train: 2 <= n_samples <= %d (got %d)", PB_SMAX, a.n_samples);
    if (!a.raw || !(a.z_vals || a.z_out)) return sw_fail(SWNERF_E_ARG, "render_pass_train: the backward needs raw and the depths (z_vals given or z_out)");
    if (a.L_pos < 0 || a.L_pos > 10 || a.L_dir < 0 || a.L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "render_pass_train: embedder bands (%d,%d) exceed (10,4)", a.L_pos, a.L_dir);
    if (a.z_vals && a.t_rand) return sw_fail(SWNERF_E_ARG, "render_pass_train: t_rand only applies to coarse sampling");
    if (a.dx) return sw_fail(SWNERF_E_ARG, "render_pass_train: no dx output (static net)");
    PassDev P;
    P.a = a;
    int rc = noview ? stream_ptrs_noview(a.packed, a.out_ch, &P.w0, &P.b0, &P.nbias, &P.two_pass)
                    : stream_ptrs(a.kind, a.packed, 0, &P.w0, &P.b0, &P.nbias, &P.two_pass);
    if (rc) return rc;
    P.act = act; P.bits = bits; P.xs = xs; P.act_d = nullptr; P.bits_d = nullptr; P.xs_d = nullptr;
    P.sort_n = 0; P.sort_s = 0;
    P.dir_steps = noview ? 0 : SW_STEPS_DIR;
    P.time_steps = 0; P.tb_off = 0;
    P.warm_steps = 0; P.warm_blocks = 0; P.skew_mode = 0; P.skew_unit = 0;
    size_t lds = PassLds<false, true>::FIXED * sizeof(float);
    if (a.n_importance > 0) {
        if (!a.z_fine) return sw_fail(SWNERF_E_ARG, "render_pass_train: n_importance>0 needs z_fine");
        if (a.n_samples < 3 || a.n_samples > SW_LDS_SC || a.n_samples + a.n_importance > SW_LDS_SORT)
            return sw_fail(SWNERF_E_UNSUPP, "render_pass_train: resampling supports 3<=N_samples<=%d and N_samples+N_importance<=%d", SW_LDS_SC, SW_LDS_SORT);
        int p2 = 2;
        while (p2 < a.n_importance) p2 <<= 1;
        P.sort_n = p2;
        p2 = 2;
        while (p2 < a.n_samples) p2 <<= 1;
        P.sort_s = p2;
        lds += 4 * SW_LDS_WAVE_FLOATS * sizeof(float);
    }
    const dim3 grid((unsigned)((a.n_rays + 3) / 4)), block(256);
    pass_startup_args(P, grid.x, noview ? SW_NOVIEW_STEPS : SW_CANON_STEPS);
    if (noview) hipLaunchKernelGGL((render_pass_kernel<false, true, 0, false>), grid, block, lds, (hipStream_t)stream, P);
    else hipLaunchKernelGGL((render_pass_kernel<false, true>), grid, block, lds, (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "render_pass_train launch");
}

extern "C" int swnerf_render_pass_backward(const float* packed_bwd, const float* bits, const float* raw, const float* z_vals,
                                           const float* ray_batch, int cols, const float* noise, int64_t n_rays, int n_samples,
                                           int white_bkgd, const float* g_rgb, const float* g_disp, const float* g_acc,
                                           const float* g_raw, float* grad, float* d_raw, void* stream) {
    if (n_rays == 0 && packed_bwd) return 0;
    if (!packed_bwd || !bits || !raw || !z_vals || !ray_batch || !grad || !d_raw || n_rays < 0)
        return sw_fail(SWNERF_E_ARG, "render_pass_backward: NULL pointer or negative n_rays");
    if (n_samples < 2 || n_samples > PB_SMAX) return sw_fail(SWNERF_E_UNSUPP, "render_pass_backward: 2 <= n_samples <= %d (got %d)", PB_SMAX, n_samples);
    if (cols < 8) return sw_fail(SWNERF_E_ARG, "render_pass_backward: ray_batch needs >= 8 columns");
    PassBwdDev P;
    P.w0 = packed_bwd; P.b0 = packed_bwd + SW_BWD_W_FLOATS; P.bits = bits; P.raw = raw; P.z = z_vals; P.ray_batch = ray_batch;
    P.cols = cols; P.noise = noise; P.n_rays = n_rays; P.S = n_samples; P.white = white_bkgd;
    P.g_rgb = g_rgb; P.g_disp = g_disp; P.g_acc = g_acc; P.g_raw = g_raw; P.grad = grad; P.d_raw = d_raw;
    const size_t lds = (SW_BWD_BIAS_TILES * SW_BIAS_TILE_FLOATS + 4 * SW_LDS_RING_FLOATS + 4 * PB_WAVE_FLOATS) * sizeof(float);
    const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
    P.bits_d = nullptr; P.dx = nullptr; P.g_pd = nullptr; P.Lp = 0; P.grad_d = nullptr; P.g_dx = nullptr;
    P.out_ch = 4;
    hipLaunchKernelGGL(render_pass_backward_kernel<0>, grid, block, lds, (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "render_pass_backward launch");
}

// ... and for the net without view directions (SWNERF_NET_NOVIEW): raw / g_raw [N,S,out_ch], d_raw [rows, 8] (columns
// >= out_ch zero: the 16-byte aligned A operand of output_linear's weight-gradient GEMM), packed_bwd = swnerf_pack_net_bwd_noview
extern "C" int swnerf_render_pass_backward_noview(const float* packed_bwd, const float* bits, const float* raw, const float* z_vals,
                                                  const float* ray_batch, int cols, const float* noise, int64_t n_rays, int n_samples,
                                                  int white_bkgd, int out_ch, const float* g_rgb, const float* g_disp, const float* g_acc,
                                                  const float* g_raw, float* grad, float* d_raw8, void* stream) {
    if (n_rays == 0 && packed_bwd) return 0;
    if (!packed_bwd || !bits || !raw || !z_vals || !ray_batch || !grad || !d_raw8 || n_rays < 0)
        return sw_fail(SWNERF_E_ARG, "render_pass_backward_noview: NULL pointer or negative n_rays");
    if (n_samples < 2 || n_samples > PB_SMAX) return sw_fail(SWNERF_E_UNSUPP, "render_pass_backward_noview: 2 <= n_samples <= %d (got %d)", PB_SMAX, n_samples);
    if (cols < 8 || out_ch < 4 || out_ch > SW_NOVIEW_MAX_OUT) return sw_fail(SWNERF_E_ARG, "render_pass_backward_noview: cols %d / out_ch %d", cols, out_ch);
    PassBwdDev P;
    P.w0 = packed_bwd; P.b0 = packed_bwd + SW_DBWD_W_FLOATS; P.bits = bits; P.raw = raw; P.z = z_vals; P.ray_batch = ray_batch;
    P.cols = cols; P.noise = noise; P.n_rays = n_rays; P.S = n_samples; P.white = white_bkgd;
    P.g_rgb = g_rgb; P.g_disp = g_disp; P.g_acc = g_acc; P.g_raw = g_raw; P.grad = grad; P.d_raw = d_raw8;
    P.bits_d = nullptr; P.dx = nullptr; P.g_pd = nullptr; P.Lp = 0; P.grad_d = nullptr; P.g_dx = nullptr;
    P.out_ch = out_ch;
    const size_t lds = (SW_NVBWD_BIAS_TILES * SW_BIAS_TILE_FLOATS + 4 * SW_LDS_RING_FLOATS + 4 * PB_WAVE_FLOATS) * sizeof(float);
    const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
    hipLaunchKernelGGL(render_pass_backward_kernel<2>, grid, block, lds, (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "render_pass_backward_noview launch");
}

// ---- the same for DirectTemporalNeRF at t != 0 (model.py:128-151; loss of d_nerf/run_dnerf.py:690-725) ----------------
extern "C" int swnerf_render_pass_train_dnerf(const swnerf_pass_args* args, float* act, float* bits, float* xs,
                                              float* act_d, float* bits_d, float* xs_d, void* stream) {
    if (!args) return sw_fail(SWNERF_E_ARG, "render_pass_train_dnerf: NULL args");
    const swnerf_pass_args& a = *args;
    if (a.n_rays == 0 && a.packed) return 0;
    if (!a.packed || !a.ray_batch || !act || !bits || !xs || !act_d || !bits_d || !xs_d) return sw_fail(SWNERF_E_ARG, "render_pass_train_dnerf: NULL pointer");
    if (a.kind != SWNERF_NET_DNERF || a.cols != 12 || !a.run_deform)
        return sw_fail(SWNERF_E_UNSUPP, "render_pass_train_dnerf: DirectTemporalNeRF with the deformation pass (t != 0) and a 12-column ray batch");
    if (a.n_rays < 0 || a.n_samples < 2 || a.n_samples > PB_SMAX) return sw_fail(SWNERF_E_UNSUPP, "render_pass_train_dnerf: 2 <= n_samples <= %d (got %d)", PB_SMAX, a.n_samples);
    if (a.n_importance != 0) return sw_fail(SWNERF_E_UNSUPP, "render_pass_train_dnerf: no resampling in the training pass (give the depths)");
    if (!a.raw || !a.dx || !(a.z_vals || a.z_out)) return sw_fail(SWNERF_E_ARG, "render_pass_train_dnerf: the backward needs raw, dx and the depths (z_vals given or z_out)");
    if (a.L_pos < 0 || a.L_pos > 10 || a.L_dir < 0 || a.L_dir > 4 || a.L_time < 0 || a.L_time > 10)
        return sw_fail(SWNERF_E_UNSUPP, "render_pass_train_dnerf: embedder bands (%d,%d,%d) exceed (10,4,10)", a.L_pos, a.L_dir, a.L_time);
    if (a.z_vals && a.t_rand) return sw_fail(SWNERF_E_ARG, "render_pass_train_dnerf: t_rand only applies to coarse sampling");
    PassDev P;
    P.a = a;
    int rc = stream_ptrs(a.kind, a.packed, 1, &P.w0, &P.b0, &P.nbias, &P.two_pass);
    if (rc) return rc;
    P.act = act; P.bits = bits; P.xs = xs; P.act_d = act_d; P.bits_d = bits_d; P.xs_d = xs_d;
    P.sort_n = 0; P.sort_s = 0;
    P.dir_steps = SW_STEPS_DIR;
    P.time_steps = SW_STEPS_TIME;
    P.tb_off = PassLds<true, true>::FIXED;       // the four waves' per-ray TIME tiles, behind everything else
    P.warm_steps = 0; P.warm_blocks = 0; P.skew_mode = 0; P.skew_unit = 0;
    const size_t lds = (PassLds<true, true>::FIXED + 4 * SW_TB_LDS_FLOATS) * sizeof(float);
    const dim3 grid((unsigned)((a.n_rays + 3) / 4)), block(256);
    pass_startup_args(P, grid.x, SW_DEFORM_STEPS + SW_CANON_STEPS);
    hipLaunchKernelGGL((render_pass_kernel<true, true>), grid, block, lds, (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "render_pass_train_dnerf launch");
}

extern "C" int swnerf_render_pass_backward_dnerf(const float* packed_bwd_fused, const float* bits, const float* bits_d, const float* raw,
                                                 const float* z_vals, const float* ray_batch, int cols, const float* noise,
                                                 const float* dx, const float* g_position_delta, int64_t n_rays, int n_samples,
                                                 int white_bkgd, int L_pos, const float* g_rgb, const float* g_disp, const float* g_acc,
                                                 const float* g_raw, float* grad, float* grad_d, float* d_raw, float* g_dx, void* stream) {
    if (n_rays == 0 && packed_bwd_fused) return 0;
    if (!packed_bwd_fused || !bits || !bits_d || !raw || !z_vals || !ray_batch || !dx || !grad || !grad_d || !d_raw || !g_dx || n_rays < 0)
        return sw_fail(SWNERF_E_ARG, "render_pass_backward_dnerf: NULL pointer or negative n_rays");
    if (n_samples < 2 || n_samples > PB_SMAX) return sw_fail(SWNERF_E_UNSUPP, "render_pass_backward_dnerf: 2 <= n_samples <= %d (got %d)", PB_SMAX, n_samples);
    if (cols < 8 || L_pos < 0 || L_pos > 10) return sw_fail(SWNERF_E_ARG, "render_pass_backward_dnerf: cols %d / L_pos %d", cols, L_pos);
    PassBwdDev P;
    P.w0 = packed_bwd_fused; P.b0 = packed_bwd_fused + SW_BWD_DN_W_FLOATS; P.bits = bits; P.raw = raw; P.z = z_vals; P.ray_batch = ray_batch;
    P.cols = cols; P.noise = noise; P.n_rays = n_rays; P.S = n_samples; P.white = white_bkgd;
    P.g_rgb = g_rgb; P.g_disp = g_disp; P.g_acc = g_acc; P.g_raw = g_raw; P.grad = grad; P.d_raw = d_raw;
    P.bits_d = bits_d; P.dx = dx; P.g_pd = g_position_delta; P.Lp = L_pos; P.grad_d = grad_d; P.g_dx = g_dx;
    const size_t lds = ((SW_BWD_BIAS_TILES + SW_DBWD_BIAS_TILES) * SW_BIAS_TILE_FLOATS + 4 * SW_LDS_RING_FLOATS + 4 * PB_WAVE_FLOATS) * sizeof(float);
    const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
    P.out_ch = 4;
    hipLaunchKernelGGL(render_pass_backward_kernel<1>, grid, block, lds, (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "render_pass_backward_dnerf launch");
}
