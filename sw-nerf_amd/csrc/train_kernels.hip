// train_kernels.hip - the training path of the MLP: the forward that saves activations
// and ReLU bit masks, the dX chains of the canonical and the deformation net.  Same building blocks as the render
// kernels (mlp_core.h) but with a 16-deep weight ring: these kernels also issue stores, `vmcnt` retires in order, and
// a store's write acknowledge takes longer than 6 ring steps often enough to cost 8 % (measured: forward 7.57 ->
// 6.94 ms, dX chain 6.64 -> 6.32 ms at 786 k rows).  The render kernels keep 8: their LDS also holds 28 KB of
// resampling scratch, and depth made no difference there.
#define SW_RING 16
#include "mlp_kernels.h"

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
    // d feature = views_linears.0.weight[:, :256]^T . d hv            (no activation on feature_linear)
    f32x16 in[8], out[8];
    seg_mfma<8, 4, SEG_ZERO, 4>(in, dhv, ws, 1.f, SideStore{grad_row + SW_ACT_HV, nullptr, nomask});
    // d h7 = feature_linear.weight^T . d feature + alpha_linear.weight * d sigma, masked by h7 > 0
    seg_mfma<8, 8, SEG_BIAS_SCALED, 8>(out, in, ws, dr[3], SideStore{grad_row + SW_ACT_FEAT, nullptr, nomask});
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
    ws_start(ws, P.w0, lds_bias, lds_ring, lane);
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
