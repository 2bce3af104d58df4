// mlp_kernels.h - what the inference (render_kernels.hip, ring depth 8) and the training (train_kernels.hip, ring
// depth 16) translation units share: the LDS layout of a 4-wave workgroup, the per-row MLP kernel template and the
// stream selection.  SW_RING must be defined (or left at its default) before this header is included.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "swnerf_common.h"
#include "mlp_core.h"
#include "host_util.h"

#define SW_LDS_BIAS_FLOATS ((SW_DEFORM_BIAS_TILES + SW_CANON_BIAS_TILES) * SW_BIAS_TILE_FLOATS)
#define SW_ZSLOT_FLOATS 128          // per wave: two 64-float slots for the next tile's depths (fine pass)
#define SW_LDS_RING_FLOATS (SW_RING * SW_STEP_FLOATS + SW_EMB_LDS_FLOATS + SW_ZSLOT_FLOATS + SW_VB_LDS_FLOATS)   // per wave: weight ring + parked embedding + depths + per-ray view-layer init tiles
#define SW_LDS_FIXED_FLOATS (SW_LDS_BIAS_FLOATS + 4 * SW_LDS_RING_FLOATS)

// ------------------------------------------------------------------------------------------
// model.forward(x) on already-embedded rows (API parity with run_network / extract_mesh):
// one wave per 32 rows; the embedded features are gathered from x into the B-operand slots.
struct MlpDev {
    const float* x; int64_t M; int C;   // C = C_pos + C_dir
    int Lp, Ld, Lt, Cpos;
    const float* t_emb; int Ct;
    const float* w0; const float* b0; int nbias; int two_pass;     // w0: the stream as stream_ptrs gives it (DIR prefix first)
    const float* wvl;       // the blob's views-loop stream: the 4 x 9 view layer on [h7 | gamma(d)] (directions vary per row here)
    float* out; float* dx;
    float* act;             // TRAIN: [M, SW_ACT_LD] activations saved for the backward pass
    float* bits;            // TRAIN: [ceil(M/32), SW_MASK_TILE_FLOATS] ReLU bit masks (mlp_core.h relu_bits)
};

template <bool DNERF, bool TRAIN = false>
__global__ void __launch_bounds__(256, 1) mlp_forward_kernel(MlpDev P) {
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
    const float* xr = P.x + (live ? row : P.M - 1) * P.C;

    f32x16 emb[2], in[8], out[8];
    float head[3], rgb[3];
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        const int col = sw_pos_col(a, h, P.Lp);
        emb[a >> 4][a & 15] = (col >= 0) ? xr[col] : 0.f;
    }
    float* act_row = TRAIN ? P.act + (live ? row : P.M - 1) * SW_ACT_LD + 4 * h : nullptr;
    float* mask_tile = TRAIN ? P.bits + tile * SW_MASK_TILE_FLOATS + lane * 4 : nullptr;
    f32x4 mb = {0.f, 0.f, 0.f, 0.f};
    WStream ws;
    // per-row directions: skip the per-ray DIR prefix of the stream (and its b_vf tiles), take the view layer from the views loop
    ws_start(ws, P.w0 + SW_STEPS_DIR * SW_STEP_FLOATS, lds_bias + SW_DIR_BIAS_TILES * SW_BIAS_TILE_FLOATS, lds_ring, lane);
    float ex = 0.f, ey = 0.f, ez = 0.f;
    if (DNERF) {
        const float ft = P.t_emb ? P.t_emb[(live ? row : P.M - 1) * P.Ct] : 0.f;   // column 0 of gamma(t) is t
#pragma nounroll
        for (int pass = P.two_pass ? 0 : 1; pass < 2; ++pass) {
            trunk_pass<true>(emb, lds_emb, ft, pass == 0, h, in, out, head, ws);
            if (pass == 0) {
                ex = head[0]; ey = head[1]; ez = head[2];
                pe_pos(xr[0] + ex, xr[1] + ey, xr[2] + ez, h, emb);      // embed_fn(input_pts_orig + dx)
            }
        }
    } else {
        trunk_pass<false, TRAIN>(emb, lds_emb, 0.f, false, h, in, out, head, ws, act_row, mask_tile, false, &mb);
    }
    // views_linears[0] on cat[feature, input_views] with feature_linear folded in (swnerf_common.h SW_CANON_STEPS): one
    // segment on [gamma(d) | h7] from the views-loop stream; the view-direction features are gathered like the position ones.
    const float* hb_rgb = ws.bias - SW_BIAS_TILE_FLOATS;      // [b_alpha, b_r, b_g, b_b]
    f32x16 demb, hv[4];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const int col = sw_dir_col(a, h, P.Ld);
        demb[a] = (col >= 0) ? xr[P.Cpos + col] : 0.f;
    }
    if (TRAIN) {                                              // h7 and its mask: stored on the spot (op path; the fused passes side-store)
        tiles_store<8>(act_row + 256 * 7, in);
        *reinterpret_cast<f32x4*>(mask_tile + 256 * 7) = mb;
    }
    ws_restart(ws, P.wvl);
    canon_tail_rows(in, demb, hv, lds_bias + h * 16, ws);
    if (TRAIN) {
        tiles_store<4>(act_row + SW_ACT_HV, hv);
        *reinterpret_cast<f32x4*>(mask_tile + 256 * 8) = relu_bits<4>(hv);
    }
    head_valu<3, 4>(hv, ws, rgb);
    rgb[0] += hb_rgb[1]; rgb[1] += hb_rgb[2]; rgb[2] += hb_rgb[3];
    if (live && h == 0) {
        f32x4 r4 = {rgb[0], rgb[1], rgb[2], head[0]};
        *reinterpret_cast<f32x4*>(P.out + row * 4) = r4;
        if (P.dx) { P.dx[row * 3 + 0] = ex; P.dx[row * 3 + 1] = ey; P.dx[row * 3 + 2] = ez; }
    }
}

// ------------------------------------------------------------------------------------------
static inline int stream_ptrs(int kind, const float* packed, int run_deform, const float** w0, const float** b0, int* nbias, int* two) {
    *nbias = SW_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS;
    if (kind == SWNERF_NET_CANON) {
        *w0 = packed; *b0 = packed + SW_CANON_W_FLOATS; *two = 0;
    } else if (kind == SWNERF_NET_DNERF) {
        if (run_deform) { *w0 = packed; *b0 = packed + SW_DNERF_W_FLOATS; *two = 1; *nbias = SW_LDS_BIAS_FLOATS; }
        else { const float* c = packed + SW_DNERF_A_FLOATS; *w0 = c; *b0 = c + SW_CANON_W_FLOATS; *two = 0; }
    } else {
        return sw_fail(SWNERF_E_ARG, "unknown net kind %d", kind);
    }
    return 0;
}

// the views-loop stream of a blob (the 4 x 9 view layer on [h7 | gamma(d)] for per-row directions): in its (last) CANON blob
static inline const float* views_loop_ptr(int kind, const float* packed) {
    return packed + (kind == SWNERF_NET_DNERF ? SW_DNERF_A_FLOATS : 0) + SW_CANON_VL_OFFSET;
}

// SWNERF_NET_NOVIEW: stream and bias tiles of the net without view directions (swnerf_pack_net_noview)
static inline int stream_ptrs_noview(const float* packed, int out_ch, const float** w0, const float** b0, int* nbias, int* two) {
    if (out_ch < 4 || out_ch > SW_NOVIEW_MAX_OUT) return sw_fail(SWNERF_E_UNSUPP, "net without view directions: out_ch %d (4 or 5)", out_ch);
    *w0 = packed; *b0 = packed + SW_NOVIEW_W_FLOATS; *two = 0;
    *nbias = SW_NOVIEW_BIAS_TILES(out_ch) * SW_BIAS_TILE_FLOATS;
    return 0;
}

