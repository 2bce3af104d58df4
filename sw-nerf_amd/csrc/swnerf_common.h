// swnerf_common.h - layout constants and small math helpers shared by the pack kernel,
// the fused MLP/render kernels and the host-side unit checks (compiles as plain C++ too).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SW_HD __host__ __device__ __forceinline__
#else
#define SW_HD static inline
#endif

// ---- packed weight stream -------------------------------------------------------------
// One "step" = the A operands of 4 consecutive v_mfma_f32_32x32x2_f32: 64 lanes x float4
// = 256 floats = 1 KiB, read by ONE global_load_dwordx4 per lane.
#define SW_STEP_FLOATS 256
#ifndef SW_RING
#define SW_RING 8                 // steps kept in flight per wave (prefetch ring); a translation unit may choose 16
#endif
#define SW_TAIL 16                // every stream ends with a copy of its first SW_TAIL steps (>= any ring depth)
#define SW_BIAS_TILE_FLOATS 32    // per 32-feature output tile: [h(2)][r(16)]

// steps per segment (NT * KT * 4)
#define SW_STEPS_EMB    64        // 8 n-tiles x 2 pos-emb k-tiles
#define SW_STEPS_EMB_T  96        // 8 x 1 (TIME: bias + time emb; once per RAY in the fused passes) + 8 x 2 (pos emb): deformation layer 0
#define SW_STEPS_TIME   32        // 8 x 1
#define SW_STEPS_TRUNK  256       // 8 x 8
#define SW_STEPS_VIEWS  144       // 4 x 9   [h7 | gamma(d)]: the view layer where directions vary per ROW (mlp_forward, point query)
#define SW_STEPS_DIR    16        // 4 x 1   gamma(d) alone: ONCE PER RAY in the fused passes (below)
#define SW_STEPS_VIEWSH 128       // 4 x 8   h7 alone: the view layer of the fused passes, initialised from the per-ray tile
// The 1- and 3-output heads (alpha_linear, rgb_linear, _time_out) are NOT in the MFMA stream: a 32-wide
// padded tile would spend 128 / 64 MFMAs on 1 / 3 useful rows.  They are VALU dot products over the
// features a lane already holds (mlp_core.h head_valu); their weights sit with the biases in LDS.
// canonical net stream: L0 | L1..L4 | L5(trunk) L5(emb) | L6 L7 | VIEWSF
// VIEWSF = feature_linear FOLDED into views_linears.0 (round 4).  feature_linear has no activation (model.py:49-51:
// feature = feature_linear(h); h = cat[feature, input_views]; views_linears[0](h)), so
//   pre_hv = Wv[:, :256] (W_f h7 + b_f) + Wv[:, 256:] gamma(d) + b_v  =  W_vf h7 + Wv[:, 256:] gamma(d) + b_vf,
//   W_vf = Wv[:, :256] . W_f  (128 x 256),   b_vf = Wv[:, :256] . b_f + b_v
// - ONE 4 x 9 segment on [h7 | gamma(d)] instead of an 8 x 8 and a 4 x 9 one: 8256 MFMAs per tile instead of 9280.  The
// pack kernel forms W_vf / b_vf with double accumulation and one rounding (pack_kernels.hip fold_views_kernel); module
// parameters, checkpoints and the weight-gradient algebra (G-based: swnerf_feature_finish) are untouched.
// The view-direction columns of the view layer are a per-RAY constant (one direction per ray, nerf/run.py:80-82 expands it over the
// samples): the fused passes evaluate  c = Wv[:, 256:] gamma(d) + b_vf  ONCE per ray (the DIR prefix of the stream: 64 MFMAs), keep
// it in a per-wave LDS tile and start every tile's view-layer accumulators from it - 8192 MFMAs per tile instead of 8256.
//   stream: DIR | MAIN = L0 | L1..L4 | L5(trunk) L5(emb) | L6 L7 | VIEWSH ;  ring tail = copy of MAIN's first SW_TAIL steps
//   (a tile rewinds to MAIN; kernels whose directions vary per row skip DIR and take the 4 x 9 view layer from the views loop)
#define SW_CANON_MAIN_STEPS (SW_STEPS_EMB + 4 * SW_STEPS_TRUNK + SW_STEPS_TRUNK + SW_STEPS_EMB + \
                             2 * SW_STEPS_TRUNK + SW_STEPS_VIEWSH)
#define SW_CANON_STEPS (SW_STEPS_DIR + SW_CANON_MAIN_STEPS)
// "bias" tiles of 32 floats ([h][r], the accumulator-init layout) in consumption order:
//   b_vf 4 (consumed by DIR, once per ray; also the init tiles of the 4 x 9 view layer) |
//   L0 8 | L1-4 32 | L5 8 | L6-7 16 | alpha_linear.weight 8, then 1 tile of head biases (alpha, r, g, b) | rgb_linear.weight 3 x 4
#define SW_CANON_BIAS_TILES (4 + 8 + 32 + 8 + 16 + 8 + 1 + 12)
#define SW_DIR_BIAS_TILES 4
#define SW_CANON_BIAS_TILE_VIEWS 0                            // index of the first views_linears (b_vf) bias tile
// the folded matrix itself rides at the end of a CANON blob: [128][SW_FOLD_LD] = [W_vf | Wv[:, 256:]] then b_vf[128]
#define SW_FOLD_LD 288
#define SW_FOLD_FLOATS (128 * SW_FOLD_LD + 128)
// deformation net stream: TIME | D0 (gamma(x) columns, accumulating) | D1..D4 | D5(trunk) D5(emb) | D6 D7 ;
//   bias tiles: 64 (the first 8 consumed by TIME) | _time_out.weight 3 x 8 | 1 head-bias tile.
// _time.0's gamma(t) columns are a per-RAY constant in the fused passes (one frame time per ray, run_dnerf.py:354-360): the D-NeRF
// blob reads DIR | TIME | MAIN = D0 .. D7 | canonical MAIN, TIME is evaluated once per ray into a per-wave LDS tile
// (mlp_core.h time_bias_tile) and every tile's D0 accumulators start from it; a tile rewinds to D0.  Kernels whose time varies
// per row (mlp_forward, the op path's training forward) run TIME and D0 in line.
#define SW_DEFORM_STEPS (SW_STEPS_EMB_T + 4 * SW_STEPS_TRUNK + SW_STEPS_TRUNK + SW_STEPS_EMB + 2 * SW_STEPS_TRUNK)
#define SW_DEFORM_BIAS_TILES (8 + 32 + 8 + 16 + 24 + 1)

// net WITHOUT view directions (use_viewdirs=False, model.py:59-60: outputs = output_linear(h)), 8x256, skip@4:
// stream L0 | L1..L4 | L5(trunk) L5(emb) | L6 L7 = 1920 steps; bias tiles: 64 | output_linear.weight out_ch x 8 | 1 tile of
// output_linear.bias.  out_ch = 4 or 5 (nerf/run.py:231: 5 when N_importance > 0; raw2outputs reads channels 0..3).
#define SW_NOVIEW_STEPS (SW_STEPS_EMB + 4 * SW_STEPS_TRUNK + SW_STEPS_TRUNK + SW_STEPS_EMB + 2 * SW_STEPS_TRUNK)
#define SW_NOVIEW_MAX_OUT 5
#define SW_NOVIEW_BIAS_TILES(out_ch) (64 + 8 * (out_ch) + 1)
#define SW_NOVIEW_W_FLOATS ((SW_NOVIEW_STEPS + SW_TAIL) * SW_STEP_FLOATS)
#define SW_NOVIEW_FLOATS (SW_NOVIEW_W_FLOATS + SW_NOVIEW_BIAS_TILES(SW_NOVIEW_MAX_OUT) * SW_BIAS_TILE_FLOATS)

// blob CANON : [DIR][canon MAIN steps][ring tail = copy of MAIN's first SW_TAIL steps][b_vf | canon bias][views loop][fold scratch]
// views loop  : [VIEWS steps][tail = copy of the first SW_TAIL VIEWS steps] - the view
//               branch as a stream that wraps onto itself, for queries of many view directions per
//               point (swnerf_query_points: trunk and density once, view branch V times)
#define SW_CANON_W_FLOATS   ((SW_CANON_STEPS + SW_TAIL) * SW_STEP_FLOATS)
#define SW_CANON_VL_OFFSET  (SW_CANON_W_FLOATS + SW_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS)
#define SW_CANON_VL_FLOATS  ((SW_STEPS_VIEWS + SW_TAIL) * SW_STEP_FLOATS)
#define SW_CANON_FOLD_OFFSET (SW_CANON_VL_OFFSET + SW_CANON_VL_FLOATS)
#define SW_CANON_FLOATS     (SW_CANON_FOLD_OFFSET + SW_FOLD_FLOATS)
// blob DNERF : [DIR][deform steps][canon MAIN steps][ring tail = head of deform][b_vf | deform bias | canon bias] then a full CANON blob
// (the CANON blob serves the `t==0 and zero_canonical` branch, model.py:143-145)
#define SW_DNERF_W_FLOATS   ((SW_DEFORM_STEPS + SW_CANON_STEPS + SW_TAIL) * SW_STEP_FLOATS)
#define SW_DNERF_A_FLOATS   (SW_DNERF_W_FLOATS + (SW_DEFORM_BIAS_TILES + SW_CANON_BIAS_TILES) * SW_BIAS_TILE_FLOATS)
#define SW_DNERF_FLOATS     (SW_DNERF_A_FLOATS + SW_CANON_FLOATS)

// ---- training path: per-row activation / gradient buffers [M, SW_ACT_LD], row-major -----------------
// columns: h_l (post-ReLU) at 256*l for l = 0..7 | 2048..2303 unused (feature_linear's output until round 3: it is never
// formed since the fold above) | views hidden (post-ReLU) at 2304
#define SW_ACT_LD    2432
#define SW_ACT_HV    2304
// backward weight stream (transposed weights, execution order of the dX chain):
// RGB^T (4x1) | VIEWSF^T = W_vf^T (8x4) | L7^T .. L1^T (8x8 each); then the alpha_linear weight as 8 bias tiles
#define SW_BWD_STEPS (16 + 128 + 7 * 256)
#define SW_BWD_W_FLOATS ((SW_BWD_STEPS + SW_TAIL) * SW_STEP_FLOATS)
#define SW_BWD_BIAS_TILES 8
// (each canonical backward blob ends with SW_FOLD_FLOATS of scratch: the folded matrix its VIEWSF^T segment was packed from)
#define SW_BWD_FLOATS (SW_BWD_W_FLOATS + SW_BWD_BIAS_TILES * SW_BIAS_TILE_FLOATS + SW_FOLD_FLOATS)
// ... with the gradient w.r.t. the embedded positions (D-NeRF training: it flows on into the deformation net):
// RGB^T | VIEWSF^T | L7^T L6^T | L5[:, :Cpos]^T (2x8) | L5[:, Cpos:]^T | L4^T .. L1^T | L0^T (2x8)
#define SW_BWD_IG_STEPS (SW_BWD_STEPS + 2 * 64)
#define SW_BWD_IG_W_FLOATS ((SW_BWD_IG_STEPS + SW_TAIL) * SW_STEP_FLOATS)
#define SW_BWD_IG_FLOATS (SW_BWD_IG_W_FLOATS + SW_BWD_BIAS_TILES * SW_BIAS_TILE_FLOATS + SW_FOLD_FLOATS)
// fused D-NeRF backward (render_pass_backward_kernel<true>): the input-gradient canonical stream, then the deformation
// stream, ONE ring; bias tiles: alpha_linear.weight (8) then _time_out.weight rows (24)
#define SW_BWD_DN_STEPS (SW_BWD_IG_STEPS + 7 * 256)
#define SW_BWD_DN_W_FLOATS ((SW_BWD_DN_STEPS + SW_TAIL) * SW_STEP_FLOATS)
#define SW_BWD_DN_FLOATS (SW_BWD_DN_W_FLOATS + (SW_BWD_BIAS_TILES + 24) * SW_BIAS_TILE_FLOATS + SW_FOLD_FLOATS)
// deformation net (`_time`): L7^T .. L1^T (trunk columns); then _time_out.weight [3,256] as 3 x 8 bias tiles
#define SW_DBWD_STEPS (7 * 256)
#define SW_DBWD_W_FLOATS ((SW_DBWD_STEPS + SW_TAIL) * SW_STEP_FLOATS)
#define SW_DBWD_BIAS_TILES 24
#define SW_DBWD_FLOATS (SW_DBWD_W_FLOATS + SW_DBWD_BIAS_TILES * SW_BIAS_TILE_FLOATS)

// net without view directions (SWNERF_NET_NOVIEW): L7^T .. L1^T like the deformation net's chain, then output_linear.weight
// [out_ch <= 5, 256] as out_ch x 8 bias-style tiles (d h7 = sum_c w_c . d raw_c on the VALU)
#define SW_NVBWD_BIAS_TILES (8 * SW_NOVIEW_MAX_OUT)
#define SW_NVBWD_FLOATS (SW_DBWD_W_FLOATS + SW_NVBWD_BIAS_TILES * SW_BIAS_TILE_FLOATS)

// ---- bf16x3 path (mlp_core_x3.h): canonical net as k-block-major groups of [A_hi 1 KiB][A_lo 1 KiB], 8 groups per chunk --
// groups: L0 8x4 | L1..L4 8x16 each | L5 8x20 (h then gamma(x)) | L6 L7 8x16 | VIEWS 4x18 (h7 through the folded W_vf, then gamma(d))
#define SW_X3_CANON_GROUPS (32 + 4 * 128 + 160 + 2 * 128 + 72)
#define SW_X3_CANON_CHUNKS (SW_X3_CANON_GROUPS / 8)
#define SW_X3_TAIL_CHUNKS 8       // the stream ends with a copy of its first chunks (>= ring slots)
#define SW_X3_CHUNK_FLOATS 4096
#define SW_X3_W_FLOATS ((SW_X3_CANON_CHUNKS + SW_X3_TAIL_CHUNKS) * SW_X3_CHUNK_FLOATS)
// blob: [weight stream + tail][the canonical bias tiles in this core's consumption order - L0 8 | L1-4 32 | L5 8 | L6-7 16 |
// alpha_linear.weight 8 + 1 head-bias tile | b_vf 4 (feature_linear folded into the view layer) | rgb_linear.weight 12;
// written by sw_pack_canon_bias_x3 (pack_kernels.hip)]
#define SW_X3_CANON_BIAS_TILES (8 + 32 + 8 + 16 + 8 + 1 + 4 + 12)
#define SW_X3_FLOATS (SW_X3_W_FLOATS + SW_X3_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS)

// D-NeRF: deformation net groups D0 8x6 (gamma(x) then gamma(t)) | D1..D4 | D5 8x20 | D6 D7, then the canonical groups
#define SW_X3_DEFORM_GROUPS (48 + 4 * 128 + 160 + 2 * 128)
#define SW_X3_DEFORM_CHUNKS (SW_X3_DEFORM_GROUPS / 8)
#define SW_X3_DNERF_W_FLOATS ((SW_X3_DEFORM_CHUNKS + SW_X3_CANON_CHUNKS + SW_X3_TAIL_CHUNKS) * SW_X3_CHUNK_FLOATS)
// blob DNERF: [deform + canon stream + tail][deform bias][canon bias] then a full CANON x3 blob (the t == 0 branch)
#define SW_X3_DNERF_A_FLOATS (SW_X3_DNERF_W_FLOATS + (SW_DEFORM_BIAS_TILES + SW_X3_CANON_BIAS_TILES) * SW_BIAS_TILE_FLOATS)
#define SW_X3_DNERF_FLOATS (SW_X3_DNERF_A_FLOATS + SW_X3_FLOATS)

// C/D register r of lane half h of v_mfma_f32_32x32x2_f32 holds row sw_frow(r,h) of the 32x32 tile
SW_HD int sw_frow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- embedding slot maps ---------------------------------------------------------------
// Which column of the reference embedding (embedder.py:33-42: [x, sin(2^0 x), cos(2^0 x), ...],
// blocks d wide) sits in B-operand slot (a, h) of an embedding k-tile; -1 = zero pad.
// Lane half h=0 evaluates the sines, h=1 the cosines of the same 16 arguments per tile.
SW_HD int sw_pos_col(int a /*0..31 over two tiles*/, int h, int L) {
    if (a < 30) { int k = a / 3, c = a % 3; return k < L ? 3 + 6 * k + 3 * h + c : -1; }
    if (a == 30) return h == 0 ? 0 : 2;
    return h == 0 ? 1 : -1;
}
SW_HD int sw_dir_col(int a /*0..15*/, int h, int L) {
    if (a < 12) { int k = a / 3, c = a % 3; return k < L ? 3 + 6 * k + 3 * h + c : -1; }
    if (a == 12) return h == 0 ? 0 : 2;
    if (a == 13) return h == 0 ? 1 : -1;
    return -1;
}
SW_HD int sw_time_col(int a /*0..15*/, int h, int L) {
    if (a < 10) return a < L ? 1 + 2 * a + h : -1;
    if (a == 10) return h == 0 ? 0 : -1;
    return -1;
}

// fused training pass: the encodings as the kernel holds them, [rows, SW_XS_LD] in B-operand SLOT order -
// k-tile kt (0,1: gamma(x); 2: gamma(d)), register r = 4g+e of lane half h sits at float 32*kt + 8g + 4h + e
// (the same map as the activation tiles above).  sw_xs_col gives the reference column of a slot (-1: zero pad).
#define SW_XS_LD 96
SW_HD int sw_xs_col(int f /*0..95*/, int Lp, int Ld) {
    const int kt = f >> 5, g = (f >> 3) & 3, h = (f >> 2) & 1, e = f & 3, r = 4 * g + e;
    return kt < 2 ? sw_pos_col(16 * kt + r, h, Lp) : sw_dir_col(r, h, Ld);
}

// ---- sin / cos -------------------------------------------------------------------------
// sin(y) for want_cos==0, cos(y) for want_cos==1, |y| up to ~1e5 (the top positional band is
// 512*x).  Three-term Cody-Waite reduction by pi/2 with FMAs, then the Cephes float minimax
// polynomials on [-pi/4, pi/4]; cos(y) = sin(y + pi/2) is a quadrant shift, so both lane
// halves run the same instruction stream.  Max abs error vs double libm ~1.2e-7 (test_host_math).
SW_HD float sw_sin_or_cos(float y, int want_cos) {
    const float n = rintf(y * 0.63661977236758134f);
    float r = fmaf(-n, 1.57079637050628662e+00f, y);
    r = fmaf(-n, -4.37113882867379223e-08f, r);
    r = fmaf(-n, -1.71512451306010346e-15f, r);
    const int q = (int)n + want_cos;
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);                                  // sin(r)
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));               // cos(r)
    float v = (q & 1) ? pc : ps;
    return (q & 2) ? -v : v;
}

// sin(y) AND cos(y) with the arithmetic of sw_sin_or_cos, bit for bit (one argument reduction and one pair of polynomials
// for both: the standalone Embedder kernel evaluates a band's sine and cosine in the same thread)
SW_HD void sw_sincos_pair(float y, float* s_out, float* c_out) {
    const float n = rintf(y * 0.63661977236758134f);
    float r = fmaf(-n, 1.57079637050628662e+00f, y);
    r = fmaf(-n, -4.37113882867379223e-08f, r);
    r = fmaf(-n, -1.71512451306010346e-15f, r);
    const int q = (int)n;
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const float sv = (q & 1) ? pc : ps, cv = ((q + 1) & 1) ? pc : ps;
    *s_out = (q & 2) ? -sv : sv;
    *c_out = ((q + 1) & 2) ? -cv : cv;
}

// torch.linspace(start, end, steps)[i] in float32 as the ATen CPU kernel computes it:
// step = (end-start)/(steps-1); i < steps/2 ? fma(step, i, start) : fma(-step, steps-1-i, end)
SW_HD float sw_linspace(float start, float end, int steps, int i) {
    if (steps <= 1) return start;                              // torch.linspace(a, b, 1) == [a]  (N_importance = 1)
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(steps - 1 - i), end);
}
