/* swnerf.h - C ABI of libswnerf_hip.so: the MI355X (gfx950) NeRF volumetric renderer.
 *
 * Drop-in boundary for the render hot path of daihangpku/SW-NeRF.
 * The reference has no FFI layer: its "operator API" is the Python symbols of ray.py,
 * embedder.py, model.py and the render_rays/run_network functions of the runner scripts.
 * Each entry point below names the reference symbol (file:line under /root/reference)
 * whose arithmetic it replaces; sw-nerf_amd/swnerf/ binds them with ctypes and
 * re-exports the reference's Python names/signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous float32 unless marked HOST;
 *     the caller (torch) allocates inputs AND outputs; nothing is retained after return
 *   - sizes are int64_t, flags int; `stream` is a hipStream_t passed as void*
 *     (torch.cuda.current_stream().cuda_stream); all work is enqueued asynchronously
 *   - return 0 on success, a negative SWNERF_E_* for argument errors, or a positive
 *     hipError_t; swnerf_last_error() returns a thread-local message
 *   - an optional output/input may be NULL where the comment says so
 */
#ifndef SWNERF_H
#define SWNERF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWNERF_VERSION 110

#define SWNERF_E_ARG      (-1)   /* bad size / NULL pointer / unsupported shape */
#define SWNERF_E_UNSUPP   (-2)   /* valid in the reference, not built here (message says what) */

/* packed-network kinds (swnerf_packed_floats / swnerf_pack_*) */
#define SWNERF_NET_CANON   0     /* vallina_NeRF == NeRFOriginal: 8x256, skip@4, view branch */
#define SWNERF_NET_DNERF   1     /* DirectTemporalNeRF: deformation net then canonical net   */
#define SWNERF_NET_NOVIEW  2     /* vallina_NeRF with use_viewdirs=False (the reference's argparse default, utils.py:43;
                                    model.py:59-60): 8x256, skip@4, outputs = output_linear(h), 4 or 5 channels */

int         swnerf_version(void);
const char* swnerf_last_error(void);

/* ---- weights -------------------------------------------------------------------------
 * The fused kernels stream weights in MFMA-fragment order (DESIGN.md "packed layout").
 * Repack after every optimizer step.  `params` is a HOST array of DEVICE pointers in
 * state_dict order of model.py:22-37 / 251-269:
 *   [0..15]  pts_linears.{0..7}.{weight,bias}   (weight [out,in] row-major as in torch)
 *   [16,17]  views_linears.0.{weight,bias}      [128, 256+C_dir]
 *   [18,19]  feature_linear.{weight,bias}       [256,256]
 *   [20,21]  alpha_linear.{weight,bias}         [1,256]
 *   [22,23]  rgb_linear.{weight,bias}           [3,128]
 * and for SWNERF_NET_DNERF additionally (model.py:108-126)
 *   [24..39] _time.{0..7}.{weight,bias}         layer 0 is [256, C_pos + C_time]
 *   [40,41]  _time_out.{weight,bias}            [3,256]
 * L_pos / L_dir / L_time = number of frequency bands of the embedders
 * (embedder.py:44-59: multires=10 -> C_pos 63, multires_views=4 -> C_dir 27,
 * time multires=10 -> C_time 21).  Limits: L_pos <= 10, L_dir <= 4, L_time <= 10.
 * Since version 109 the pack step FOLDS feature_linear into views_linears.0 (model.py:49-53: feature_linear has no
 * activation, so views_linears.0(cat[feature_linear(h), dirs]) = [Wv[:, :256] W_f | Wv[:, 256:]] . cat[h, dirs] + (Wv[:, :256] b_f
 * + b_v); the product is formed in double and rounded once): the kernels run ONE 128 x (256 + C_dir) layer where the
 * reference runs a 256 x 256 and a 128 x (256 + C_dir) one - 11 % fewer MFMAs per row, same function.  The caller's
 * tensors, their gradients (swnerf_feature_finish) and checkpoints are untouched.
 * Since version 110 a SWNERF_NET_DNERF blob carries layer 0 of the deformation net (model.py:129: _time.0 on cat[gamma(x),
 * gamma(t)]) as two segments - bias + the gamma(t) columns, then the gamma(x) columns: the fused passes evaluate the first ONCE PER
 * RAY (one frame time per ray, run_dnerf.py:354-360), the per-row entry points (swnerf_mlp_forward, deform_forward_train) in line.
 * Same function, same blob size; blobs are not interchangeable across versions (they are made per process, never stored). */
size_t swnerf_packed_floats(int kind);
int swnerf_pack_net(int kind, const float* const* params /*HOST*/, int L_pos, int L_dir,
                    int L_time, float* packed, void* stream);
/* SWNERF_NET_NOVIEW (model.py:22-37 with use_viewdirs=False): params =
 *   [0..15]  pts_linears.{0..7}.{weight,bias}
 *   [16,17]  output_linear.{weight,bias}        [out_ch,256], out_ch = 4 or 5 (nerf/run.py:231)
 * packed: swnerf_packed_floats(SWNERF_NET_NOVIEW) floats. */
int swnerf_pack_net_noview(const float* const* params /*HOST*/, int L_pos, int out_ch, float* packed, void* stream);
/* vallina_NeRF.forward for that net (model.py:39-47, 59-60) on embedded rows: x [M, ldx] whose first 3(1+2 L_pos) columns are
 * gamma(x) -> out [M, out_ch] = output_linear(h).  The op-by-op form (run_network, nerf/run.py:73-87); the fused render pass
 * takes the same packed blob with kind SWNERF_NET_NOVIEW. */
int swnerf_mlp_forward_noview(const float* packed, const float* x, int64_t M, int ldx, int L_pos, int out_ch, float* out, void* stream);

/* ---- ray.py -------------------------------------------------------------------------- */

/* get_rays (ray.py:10-38) for the pixel range [ray0, ray0+n) of an HxW image in row-major
 * order.  focal_branch!=0 selects the float-focal branch (:26-29: cx=W/2, cy=H/2, fy=fx).
 * c2w: HOST, 12 floats, row-major [3,4].  rays_o may be NULL. */
int swnerf_get_rays(int H, int W, double fx, double fy, double cx, double cy, int focal_branch,
                    const float* c2w /*HOST*/, int64_t ray0, int64_t n,
                    float* rays_o /*[n,3]*/, float* rays_d /*[n,3]*/, void* stream);

/* ndc_rays (ray.py:75-92).  In-place allowed (o_out==rays_o, d_out==rays_d). */
int swnerf_ndc_rays(int H, int W, double focal, double near, const float* rays_o, const float* rays_d,
                    int64_t n, float* o_out, float* d_out, void* stream);

/* The ray-batch packing inside render() (nerf/run.py:137-158, d_nerf/run_dnerf.py:137-160):
 * viewdirs = d/|d| taken BEFORE the optional NDC warp; rows = [o(3) d(3) near far (t) viewdirs(3)].
 * has_time!=0 -> 12 columns with frame_time at column 8, else 11 columns. */
int swnerf_pack_ray_batch(const float* rays_o, const float* rays_d, int64_t n, double near, double far,
                          int has_time, double frame_time, int ndc, int H, int W, double ndc_focal,
                          float* ray_batch, void* stream);

/* raw2outputs (ray.py:155-198).  noise: NULL or [N,S] already multiplied by raw_noise_std.
 * Any output may be NULL.  disp is NaN where acc==0, like the reference. */
int swnerf_raw2outputs(const float* raw /*[N,S,4]*/, const float* z_vals /*[N,S]*/,
                       const float* rays_d /*[N,3]*/, const float* noise, int64_t N, int S,
                       int white_bkgd, float* rgb_map /*[N,3]*/, float* disp_map, float* acc_map,
                       float* weights /*[N,S]*/, float* depth_map, void* stream);

/* Gradient of raw2outputs w.r.t. raw (autograd of ray.py:155-198; the loss of nerf/run.py:688-697
 * reaches it through rgb_map / rgb0).  g_*: upstream gradients of the five outputs, each may be NULL;
 * d_raw [N,S,4] is overwritten.  z_vals and rays_d get no gradient (they are detached inputs of the
 * render path).  2 <= S <= 1024. */
int swnerf_raw2outputs_backward(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                                int64_t N, int S, int white_bkgd, const float* g_rgb /*[N,3]*/,
                                const float* g_disp /*[N]*/, const float* g_acc /*[N]*/, const float* g_depth /*[N]*/,
                                const float* g_weights /*[N,S]*/, float* d_raw /*[N,S,4]*/, void* stream);

/* sample_pdf (ray.py:96-153).  bins [N,nb], weights [N,nb-1]; u: NULL -> det linspace(0,1,n_samples)
 * (det=True), else [N,n_samples] uniforms (replaces torch.rand).  samples [N,n_samples].
 * If z_vals ([N,S]) and z_sorted ([N,S+n_samples]) are given, also writes
 * sort(cat[z_vals, samples]) (nerf/run.py:400); z_std ([N], std of samples, :416) may be NULL. */
int swnerf_sample_pdf(const float* bins, const float* weights, int64_t N, int nb, int n_samples,
                      const float* u, float* samples, const float* z_vals, int S, float* z_sorted,
                      float* z_std, void* stream);

/* ---- embedder.py --------------------------------------------------------------------- */

/* Embedder.embed (embedder.py:33-42): x [M,d] -> [M, d*(1+2L)], frequency-major, sin before cos. */
int swnerf_embed(const float* x, int64_t M, int d, int L, float* out, void* stream);

/* ---- model.py ------------------------------------------------------------------------ */

/* vallina_NeRF.forward / NeRFOriginal.forward (model.py:39-62, 273-296), use_viewdirs=True:
 * x [M, C_pos+C_dir] already-embedded rows -> out [M,4] = [rgb(3), sigma].
 * DirectTemporalNeRF.forward (model.py:138-151): packed kind DNERF, t_emb [M,C_time] = embedded
 * frame time (ts[0]); run_deform==0 takes the `t==0 and zero_canonical` branch (dx=0);
 * dx_out [M,3] may be NULL. */
int swnerf_mlp_forward(int kind, const float* packed, const float* x, int64_t M, int L_pos, int L_dir,
                       const float* t_emb, int L_time, int run_deform,
                       float* out /*[M,4]*/, float* dx_out /*[M,3]*/, void* stream);

/* ---- training path of the MLP (autograd of model.py:39-62) -----------------
 * forward_train: as swnerf_mlp_forward (SWNERF_NET_CANON) and additionally saves, per row, the
 *   activations the weight-gradient GEMMs need: act [M, swnerf_act_floats_per_row()] row-major
 *   (h_l post-ReLU at column 256*l, l=0..7; columns 2048..2303 unused - feature_linear's output is never formed, see
 *   swnerf_pack_net; views hidden at 2304), and the
 *   ReLU bit masks of every 32-row tile: bits [swnerf_mask_floats(M)] (1 KiB per tile and layer; opaque,
 *   only the backward_dx entry points read it).
 * pack_net_bwd: the transposed weight stream of the dX chain (params as for swnerf_pack_net, first 24).
 * backward_dx: bits from forward_train, d_out [M,4] = d raw -> grad [M, same layout as act] =
 *   d(pre-activation) of every layer.
 * gemm_tn: C[No,ldc] += A[M,lda]^T . B[M,ldb] (first No / Ni columns), bias[No] += column sums of A
 *   (bias may be NULL): dW and db of one Linear layer from `grad` and `act`/inputs.  C and bias accumulate:
 *   zero them first.  No <= 256. */
size_t swnerf_packed_bwd_floats(void);
size_t swnerf_act_floats_per_row(void);
size_t swnerf_mask_floats(int64_t M);
int swnerf_mlp_forward_train(const float* packed, const float* x, int64_t M, int L_pos, int L_dir,
                             float* out /*[M,4]*/, float* act, float* bits, void* stream);
int swnerf_pack_net_bwd(const float* const* params /*HOST*/, int L_pos, int L_dir, float* packed_bwd, void* stream);
int swnerf_mlp_backward_dx(const float* packed_bwd, const float* bits, const float* d_out /*[M,4]*/, int64_t M,
                           float* grad, void* stream);
int swnerf_gemm_tn(const float* A, int lda, int No, const float* B, int ldb, int Ni, int64_t M,
                   float* C, int ldc, float* bias, void* stream);
/* gemm_tn for a 256 x 256 block (No = Ni = 256) with up to two riders that share one of its operands and its pass
 * over the rows (each may be NULL):
 *   B2 [M, Ni2 <= 64]:  C2[256, ldc2] += A^T . B2            (the gamma(x) columns of a skip layer: same A)
 *   A2 [M, No2 <= 32]:  C3[No2, ldc3] += A2^T . B,  bias3[No2] += column sums of A2   (alpha_linear: same B as feature_linear)
 * Equivalent to the corresponding separate swnerf_gemm_tn calls (which it falls back to for small or unaligned M). */
int swnerf_gemm_tn_fused(const float* A, int lda, const float* B, int ldb, int64_t M, float* C, int ldc, float* bias,
                         const float* B2, int ldb2, int Ni2, float* C2, int ldc2,
                         const float* A2, int lda2, int No2, float* C3, int ldc3, float* bias3, void* stream);
/* Several swnerf_gemm_tn_fused problems over the SAME M rows (the weight-gradient GEMMs of one row chunk of a training step:
 * loss.backward() of nerf/run.py:700) as ONE launch: the workgroups are dealt out over the items in proportion to their work,
 * so the chunk pays one launch ramp and one atomic epilogue instead of one per layer.  Field meaning as the arguments of
 * swnerf_gemm_tn_fused (riders may be NULL).  Results equal the separate calls' (split-K atomics add in a different order). */
typedef struct swnerf_gemm_item {
    const float* A; int lda; const float* B; int ldb; float* C; int ldc; float* bias;
    const float* B2; int ldb2; int Ni2; float* C2; int ldc2;
    const float* A2; int lda2; int No2; float* C3; int ldc3; float* bias3;
} swnerf_gemm_item;
int swnerf_gemm_tn_group(const swnerf_gemm_item* items /*HOST*/, int n_items, int64_t M, void* stream);

/* ---- training path of DirectTemporalNeRF (autograd of model.py:128-151; the loss of
 * d_nerf/run_dnerf.py:690-725 needs d/d(position_delta) too).  The forward is the composition the
 * reference runs: deformation net -> dx; gamma(x + dx) (swnerf_embed); canonical net
 * (swnerf_mlp_forward_train on [gamma(x+dx), gamma(d)]).  Backward: canonical dX chain that also returns
 * the gradient w.r.t. the re-embedded positions (through the sin/cos of gamma), then the deformation dX chain
 * seeded with d dx = d pts + d position_delta.
 * Backward stream kinds for swnerf_packed_bwd_floats_kind / swnerf_pack_net_bwd_kind:
 *   SWNERF_BWD_CANON            what swnerf_pack_net_bwd writes (params: the 24 canonical tensors)
 *   SWNERF_BWD_CANON_INPUT_GRAD the same plus the position-embedding columns of pts_linears.0/.5
 *   SWNERF_BWD_DEFORM           `_time.1..7` trunk columns + `_time_out.weight` (params: the 18 tensors
 *                               _time.0.weight, _time.0.bias, ..., _time_out.weight, _time_out.bias)
 * deform_forward_train: packed = a SWNERF_NET_DNERF blob; x [M,C] as for swnerf_mlp_forward, t_emb [M,1+2*L_time];
 *   writes dx [M,3], act_d [M, swnerf_act_floats_per_row()] (h_l of `_time` at column 256*l) and
 *   bits_d [swnerf_mask_floats(M)].
 * backward_dx_pts: as swnerf_mlp_backward_dx with a SWNERF_BWD_CANON_INPUT_GRAD stream; pts [M,3] = the
 *   positions that were embedded (x + dx); also writes d_pts [M,3].
 * deform_backward_dx: bits_d, d_dx [M,3] -> grad_d [M, same layout as act_d] = d(pre-activation) of `_time.l`. */
#define SWNERF_BWD_CANON 0
#define SWNERF_BWD_CANON_INPUT_GRAD 1
#define SWNERF_BWD_DEFORM 2
#define SWNERF_BWD_DNERF_FUSED       3   /* CANON_INPUT_GRAD then DEFORM as ONE stream (params: the 42 DirectTemporalNeRF tensors) */
size_t swnerf_packed_bwd_floats_kind(int bwd_kind);
int swnerf_pack_net_bwd_kind(int bwd_kind, const float* const* params /*HOST*/, int L_pos, int L_dir,
                             float* packed_bwd, void* stream);
int swnerf_deform_forward_train(const float* packed, const float* x, const float* t_emb, int64_t M,
                                int L_pos, int L_dir, int L_time, float* dx /*[M,3]*/, float* act_d, float* bits_d, void* stream);
int swnerf_mlp_backward_dx_pts(const float* packed_bwd, const float* bits, const float* d_out /*[M,4]*/,
                               const float* pts /*[M,3]*/, int64_t M, int L_pos,
                               float* grad, float* d_pts /*[M,3]*/, void* stream);
int swnerf_deform_backward_dx(const float* packed_bwd, const float* bits_d, const float* d_dx /*[M,3]*/, int64_t M,
                              float* grad_d, void* stream);

/* network_query_fn on bare points (nerf/load_model.py:56-74; nerf/extract_mesh.py:27-90, :155-175):
 * pts [M,3] world positions, packed = a SWNERF_NET_CANON blob; the positional encodings are
 * evaluated in registers.  shared_dirs == 0: dirs [M,3], one direction per point -> out [M,4] = raw
 * [rgb(3), sigma].  shared_dirs != 0: dirs [V,3] shared by every point -> out [M,4] =
 * [mean over the V directions of the raw rgb, sigma] (what sample_grid averages); the trunk and
 * the density are evaluated once per point, only the view branch V times. */
int swnerf_query_points(const float* packed, const float* pts, int64_t M, const float* dirs, int64_t n_dirs,
                        int shared_dirs, int L_pos, int L_dir, float* out /*[M,4]*/, void* stream);

/* Coarse sampling of render_rays on its own (nerf/run.py:355-385): z_vals [N,S] = near(1-t)+far*t with t = linspace(0,1,S)
 * (or the lindisp form, :365), stratified jitter when t_rand [N,S] is given (:369-383: replaces torch.rand);
 * pts [N,S,3] = rays_o + rays_d * z (:385) may be NULL.  ray_batch as for swnerf_render_pass (columns 0-7 are read). */
int swnerf_sample_coarse(const float* ray_batch, int64_t n_rays, int cols, int n_samples, int lindisp,
                         const float* t_rand, float* z_vals /*[N,S]*/, float* pts /*[N,S,3]*/, void* stream);

/* ---- fused render pass (render_rays, nerf/run.py:316-422, d_nerf/run_dnerf.py:354-480) -------
 * One wavefront owns one ray: sampling -> positional encoding -> MLP (MFMA, register
 * resident) -> alpha compositing -> optional hierarchical resampling, with no HBM traffic
 * for pts / embeddings / activations / raw. */
typedef struct swnerf_pass_args {
    /* inputs */
    const float* ray_batch;   /* [N, cols]  cols = 11, or 12 with frame_time at column 8; SWNERF_NET_NOVIEW: 8 = [o, d, near, far]
                                 (rays without view directions, nerf/run.py:152-157) */
    int64_t      n_rays;
    int          cols;
    int          kind;        /* SWNERF_NET_CANON / SWNERF_NET_DNERF / SWNERF_NET_NOVIEW */
    const float* packed;      /* packed net of that kind */
    int          run_deform;  /* DNERF only: 0 = `t==0 and zero_canonical` branch */
    int          L_pos, L_dir, L_time;
    int          n_samples;   /* S of THIS pass */
    const float* z_vals;      /* NULL: coarse sampling from near/far (nerf/run.py:361-367);
                                 else [N,S] given depths (fine pass, or run_dnerf.py:408) */
    int          lindisp;
    const float* t_rand;      /* NULL or [N,S] stratified jitter (perturb>0, nerf/run.py:369-383) */
    const float* noise;       /* NULL or [N,S] density noise, pre-scaled (ray.py:176-184) */
    int          white_bkgd;
    /* per-ray outputs, any may be NULL */
    float* rgb_map;           /* [N,3] */
    float* disp_map;          /* [N]   */
    float* acc_map;           /* [N]   */
    float* depth_map;         /* [N]   */
    /* per-sample outputs, any may be NULL */
    float* weights;           /* [N,S]   */
    float* raw;               /* [N,S,4]  (SWNERF_NET_NOVIEW: [N,S,out_ch]) */
    float* dx;                /* [N,S,3] position_delta (DNERF) */
    float* z_out;             /* [N,S]   the depths this pass sampled */
    /* hierarchical resampling after compositing (nerf/run.py:394-400), n_importance==0: off */
    int          n_importance;
    const float* u;           /* NULL: det (perturb==0); else [N,n_importance] uniforms */
    float* z_fine;            /* [N, S+n_importance] sorted union */
    float* z_std;             /* [N] std of the new samples, may be NULL */
    int          out_ch;      /* SWNERF_NET_NOVIEW: channels of output_linear (4 or 5); ignored otherwise */
} swnerf_pass_args;

int swnerf_render_pass(const swnerf_pass_args* args /*HOST*/, void* stream);

/* ---- opt-in reduced-cost precision: "bf16x3" ----------------------------------------------------------------
 * swnerf_render_pass with the MLPs on the bf16 matrix pipe: every fp32 weight and activation is split into two bf16
 * halves (hi + lo, 16 significant bits) and each product is three v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 * (W_hi.x_hi + W_hi.x_lo + W_lo.x_hi).  Sampling, encodings, heads, compositing and resampling stay fp32 and are the
 * same code as swnerf_render_pass.  NOT the parity path (that is fp32 MFMA, bit-comparable to torch's CPU kernels up to
 * summation order); measured deviation from it: DESIGN.md 7c.  Inference only; both net kinds.
 * args->packed = a blob from swnerf_pack_net_x3_kind (swnerf_packed_x3_floats_kind(kind) floats), built from the same
 * tensors plus the fp32 blob of swnerf_pack_net for that kind (its bias tiles are copied).
 * terms: 3 = bf16x3, 1 = plain bf16 (hi halves only; a yardstick, ~37 dB).
 * swnerf_packed_x3_floats / swnerf_pack_net_x3: the SWNERF_NET_CANON forms. */
size_t swnerf_packed_x3_floats_kind(int kind);
int swnerf_pack_net_x3_kind(int kind, const float* const* params /*HOST*/, int L_pos, int L_dir, int L_time,
                            const float* packed_fp32, float* packed_x3, void* stream);
size_t swnerf_packed_x3_floats(void);
int swnerf_pack_net_x3(const float* const* params /*HOST*/, int L_pos, int L_dir, const float* packed_canon,
                       float* packed_x3, void* stream);
int swnerf_render_pass_x3(const swnerf_pass_args* args /*HOST*/, int terms, void* stream);

/* ---- the fused pass under autograd: loss.backward() of the reference's training step --------------------------
 * (nerf/run.py:684-708: render -> img2mse(rgb) + img2mse(rgb0) -> backward -> optimizer.step)
 * Static net (SWNERF_NET_CANON, 11-column ray batch), 2 <= n_samples <= 256.
 *
 * render_pass_train: exactly swnerf_render_pass (same sampling / encoding / MLP / compositing / resampling
 * arithmetic, same outputs) and additionally saves what the backward needs.  Rows of the saved buffers are the
 * (ray, sample) rows PADDED to whole 32-sample tiles per ray: rows = swnerf_train_rows(N, S) = N * ceil(S/32) * 32,
 * row = (ray * ceil(S/32) + s/32) * 32 + s%32.
 *   act  [rows, swnerf_act_floats_per_row()]   post-ReLU activations (as swnerf_mlp_forward_train)
 *   bits [swnerf_mask_floats(rows)]            ReLU bit masks
 *   xs   [rows, swnerf_xs_floats_per_row()]    gamma(x) (64 slots) and gamma(d) (32 slots) in the kernel's operand
 *                                              slot order; swnerf_unslot_grad maps slot columns back
 * args->raw and the depths (args->z_vals given, or args->z_out) are required: the backward recomputes the
 * compositing from them.
 *
 * render_pass_backward: gradients of (rgb_map, disp_map, acc_map) -> d raw [rows,4] (padded rows; zeros past S) and
 * the gradient of every layer's pre-activation grad [rows, act floats] (as swnerf_mlp_backward_dx), one wavefront
 * per ray: compositing backward in the wave's LDS slice, then the dX chain tile by tile.  packed_bwd:
 * swnerf_pack_net_bwd_kind(SWNERF_BWD_CANON).  g_* may each be NULL; g_raw = upstream gradient of the returned raw
 * (retraw=True: the reference's trainers ask for it, nerf/run.py:685) is added to d raw.  z_vals [N,S]: the depths the forward used. */
int64_t swnerf_train_rows(int64_t n_rays, int n_samples);
int swnerf_xs_floats_per_row(void);
int swnerf_render_pass_train(const swnerf_pass_args* args /*HOST*/, float* act, float* bits, float* xs, void* stream);
int swnerf_render_pass_backward(const float* packed_bwd, const float* bits, const float* raw /*[N,S,4]*/,
                                const float* z_vals /*[N,S]*/, const float* ray_batch, int cols, const float* noise,
                                int64_t n_rays, int n_samples, int white_bkgd, const float* g_rgb /*[N,3]*/,
                                const float* g_disp /*[N]*/, const float* g_acc /*[N]*/, const float* g_raw /*[N,S,4] or NULL*/,
                                float* grad, float* d_raw, void* stream);
/* The same pair for the net WITHOUT view directions (SWNERF_NET_NOVIEW; use_viewdirs=False is the reference's argparse
 * default, nerf/run.py:461, and its create_nerf then builds output_ch = 5 when N_importance > 0, nerf/run.py:231):
 * swnerf_render_pass_train takes kind SWNERF_NET_NOVIEW with an 8-column ray batch (raw is [N,S,out_ch]; act holds h0..h7
 * in its first 2048 columns; xs gamma(x) in its first 64 slots), and this backward is its counterpart: packed_bwd =
 * swnerf_pack_net_bwd_noview (pts_linears.7..1 transposed + output_linear.weight), raw / g_raw [N,S,out_ch], and d_raw8
 * [rows, 8] = d raw in columns 0..out_ch-1, zeros behind - the 16-byte aligned A operand of output_linear's weight-
 * gradient GEMM (swnerf_gemm_tn with No = 8).  grad: d pre-activation of pts_linears.0..7 in columns 0..2047. */
size_t swnerf_packed_bwd_noview_floats(void);
int swnerf_pack_net_bwd_noview(const float* const* params /*HOST; as swnerf_pack_net_noview*/, int L_pos, int out_ch,
                               float* packed_bwd, void* stream);
int swnerf_render_pass_backward_noview(const float* packed_bwd, const float* bits, const float* raw /*[N,S,out_ch]*/,
                                       const float* z_vals /*[N,S]*/, const float* ray_batch, int cols, const float* noise,
                                       int64_t n_rays, int n_samples, int white_bkgd, int out_ch, const float* g_rgb /*[N,3]*/,
                                       const float* g_disp /*[N]*/, const float* g_acc /*[N]*/,
                                       const float* g_raw /*[N,S,out_ch] or NULL*/, float* grad, float* d_raw8, void* stream);
/* The same pair for DirectTemporalNeRF at t != 0 (model.py:128-151; the loss of d_nerf/run_dnerf.py:690-725 puts
 * gradients on the image AND on position_delta).  No resampling in the training pass (n_importance must be 0: the
 * one-model configuration's coarse pass is a no_grad inference pass, run_dnerf.py:417-421).  args->dx (position_delta
 * [N,S,3]) and args->raw are required outputs.  *_d: the deformation net's buffers, sized like the canonical ones
 * (act_d uses the first 2048 columns; xs_d = gamma(x) 64 slots + gamma(t) 32 slots).
 * backward: packed_bwd_fused = swnerf_pack_net_bwd_kind(SWNERF_BWD_DNERF_FUSED); dx = the forward's position_delta;
 * g_position_delta [N,S,3] its upstream gradient or NULL; outputs grad / grad_d [rows, act floats], d_raw [rows,4],
 * g_dx [rows,4] = d dx (4th column 0) - the A operand of the `_time_out` weight-gradient GEMM. */
int swnerf_render_pass_train_dnerf(const swnerf_pass_args* args /*HOST*/, float* act, float* bits, float* xs,
                                   float* act_d, float* bits_d, float* xs_d, void* stream);
int swnerf_render_pass_backward_dnerf(const float* packed_bwd_fused, const float* bits, const float* bits_d, const float* raw,
                                      const float* z_vals, const float* ray_batch, int cols, const float* noise, const float* dx,
                                      const float* g_position_delta, int64_t n_rays, int n_samples, int white_bkgd, int L_pos,
                                      const float* g_rgb, const float* g_disp, const float* g_acc, const float* g_raw,
                                      float* grad, float* grad_d, float* d_raw, float* g_dx, void* stream);
/* Cs [rows_w, nslots] holds weight-gradient columns in slot order (a TN GEMM against xs[:, slot0 : slot0+nslots]):
 * W[o][col0 + column(slot0 + f)] = Cs[o][f] for every real slot f; pad slots are dropped. */
int swnerf_unslot_grad(const float* Cs, int ld_s, int rows_w, int slot0, int nslots, int L_pos, int L_dir,
                       float* W, int ldw, int col0, void* stream);
/* No training pass (fused or op by op) stores `feature` or d feature or runs feature_linear's weight-gradient GEMM
 * (feature_linear has no activation, model.py:50-51; the forward and dX kernels run it folded into the view layer).  From G [128,256] = sum_rows d pre_hv (x) h7 (swnerf_gemm_tn of the
 * gradient rows' view-hidden columns against h7), db_hv [128] (its bias output) and the CURRENT weights this adds
 *   dWv[u][o] += sum_i G[u][i] W_f[o][i] + db_hv[u] b_f[o]      (d views_linears.0.weight[:, :256]; Wv/dWv: [128, ld >= 256])
 *   dW_f[o][i] += sum_u Wv[u][o] G[u][i],   db_f[o] += sum_u Wv[u][o] db_hv[u]      (d feature_linear.weight / .bias)
 *   dW_alpha[i] += a4w[3][i],  db_alpha += a4b[3]      (alpha_linear from the 4-row form: a4w [4,256] = d raw^T . h7, a4b [4]) */
int swnerf_feature_finish(const float* G, const float* db_hv, const float* Wv, int ldwv, const float* W_f, const float* b_f,
                          const float* a4w, const float* a4b, float* dWv, int ld_dwv, float* dW_f, float* db_f,
                          float* dW_alpha, float* db_alpha, void* stream);
/* The five narrow weight-gradient products of the canonical net's fused training pass in ONE pass over M rows (instead of five
 * swnerf_gemm_tn calls that read h7, d pre_hv and the xs rows twice each).  grad / act: the fused pass's gradient and activation
 * rows [M, ld >= 2432] (columns: d pre_0 0..255, d pre_hv 2304..2431; h7 1792..2047, hv 2304..2431), xs [M, 96], d_out [M, 4].
 * Accumulates (+=): c0s [256,64] = d pre_0^T xs[:, :64];  cvs [128,32] = d pre_hv^T xs[:, 64:96];  G [128,256] = d pre_hv^T h7;
 * a4w [4,256] = d_out^T h7;  rgb4 [4,128] = d_out^T hv;  b_l0 [256], b_hv [128], a4b [4], rgb4b [4] = the column sums of d pre_0,
 * d pre_hv and d_out (each may be NULL). */
int swnerf_canon_narrow_grads(const float* grad, int ldg, const float* act, int lda, const float* xs, const float* d_out, int64_t M,
                              float* c0s, float* cvs, float* G, float* a4w, float* rgb4, float* b_l0, float* b_hv, float* a4b,
                              float* rgb4b, void* stream);
/* The same for the other two nets (one launch per chunk each; table-driven narrow_plan_kernel, csrc/backward_kernels.hip):
 * deformation net (`_time.0` = [gamma(x) | gamma(t)], `_time_out`; model.py:128-136): grad_d / act_d [M, ld >= 2432] (d pre_0 at
 *   column 0, h7 at 1792), xs_d [M, 96] (gamma(x) slots 0..63, gamma(t) slots 64..95), g_dx [M, 4] = d dx with a zero 4th column:
 *   c0s [256,64] += d pre_0^T xs_d[:, :64], cts [256,32] += d pre_0^T xs_d[:, 64:], w4 [4,256] += g_dx^T h7; b_l0 [256], b4 [4]
 * net without view directions (pts_linears.0, output_linear; model.py:59-60): xs [M, 96], d_raw8 [M, 8] (columns >= out_ch zero):
 *   c0s [256,64] += d pre_0^T xs[:, :64], w8 [8,256] += d_raw8^T h7; b_l0 [256], b8 [8].
 * All operands 16-byte aligned, leading dimensions multiples of 4; outputs accumulate (atomics). */
int swnerf_deform_narrow_grads(const float* grad_d, int ldg, const float* act_d, int lda, const float* xs_d, const float* g_dx, int64_t M,
                               float* c0s, float* cts, float* w4, float* b_l0, float* b4, void* stream);
int swnerf_noview_narrow_grads(const float* grad, int ldg, const float* act, int lda, const float* xs, const float* d_raw8, int64_t M,
                               float* c0s, float* w8, float* b_l0, float* b8, void* stream);
/* ... for xs_d: slots 64..95 hold gamma(t) (L_time bands) instead of gamma(d) */
int swnerf_unslot_grad_time(const float* Cs, int ld_s, int rows_w, int nslots, int L_time, float* W, int ldw, int col0, void* stream);

/* ---- any-shape MLP layers (model.py:10-62, 93-151, 227-296 at shapes the fused kernels are not built for:
 * use_viewdirs=False - the reference's argparse default, utils.py:26-29 / model.py:59-60 -, other D / W / skips) -------
 * linear   : y[M,N] = act(x[M,K] . weight[N,K]^T + bias)   torch.nn.functional.linear (+ relu if relu != 0); bias may be NULL
 * gemm_nn  : c[M,N] = a[M,K] . b[K,N]                       input gradient of a linear layer: dX = dY . weight
 * relu_mask: dy[e] = y[e] > 0 ? dy[e] : 0, in place        relu backward
 * The weight gradient is swnerf_gemm_tn.  Any leading dimensions >= the row length; fp32 MFMA, fp32 accumulate. */
int swnerf_linear(const float* x, int ldx, int64_t M, int K, const float* weight /*[N,K]*/, const float* bias /*[N]*/,
                  int N, int relu, float* y, int ldy, void* stream);
int swnerf_gemm_nn(const float* a, int lda, int64_t M, int K, const float* b, int ldb, int N, float* c, int ldc,
                   void* stream);
int swnerf_relu_mask(float* dy, const float* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
