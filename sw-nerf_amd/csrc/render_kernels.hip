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

#include "render_pass.h"

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
    // directions vary per point / per loop turn: skip the stream's per-ray DIR prefix (and its b_vf tiles)
    ws_start(ws, P.w0 + SW_STEPS_DIR * SW_STEP_FLOATS, lds_bias + SW_DIR_BIAS_TILES * SW_BIAS_TILE_FLOATS, lds_ring, lane);
    trunk_pass<false>(emb, lds_emb, 0.f, false, h, in, out, head, ws);
    const float* hb_rgb = ws.bias - SW_BIAS_TILE_FLOATS;      // [b_alpha, b_r, b_g, b_b]
    const float* rgb_tiles = ws.bias;                         // rgb_linear.weight as bias-style tiles: re-read on every turn
    // feature_linear is folded into the view layer (swnerf_common.h SW_CANON_STEPS): the loop below runs the 4 x 9 segment on
    // [gamma(d) | h7] from the views-loop stream, whose tail is its own head.
    ws_restart(ws, P.wvl);
    float sr = 0.f, sg = 0.f, sb = 0.f;
    const int64_t nv = P.shared ? P.V : 1;
#pragma nounroll
    for (int64_t v = 0; v < nv; ++v) {
        const float* dp = P.dirs + (P.shared ? v : rr) * 3;
        f32x16 demb, hv[4];
        pe_dir(dp[0], dp[1], dp[2], h, demb);
        ws.bias = rgb_tiles;
        ws.base = reinterpret_cast<const char*>(P.wvl);
        canon_tail_rows(in, demb, hv, lds_bias + h * 16, ws);
        float c3[3];
        head_valu<3, 4>(hv, ws, c3);
        sr += c3[0] + hb_rgb[1]; sg += c3[1] + hb_rgb[2]; sb += c3[2] + hb_rgb[3];
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
    const bool noview = a.kind == SWNERF_NET_NOVIEW;
    if (noview ? a.cols != 8 : (a.cols != 11 && a.cols != 12))
        return sw_fail(SWNERF_E_ARG, "render_pass: ray_batch must have 11 or 12 columns (use_viewdirs) or 8 (SWNERF_NET_NOVIEW), got %d for kind %d", a.cols, a.kind);
    if (a.kind == SWNERF_NET_DNERF && a.cols != 12) return sw_fail(SWNERF_E_ARG, "render_pass: D-NeRF needs the frame_time column");
    if (noview && a.dx) return sw_fail(SWNERF_E_ARG, "render_pass: a static net has no position_delta output here");
    if (a.L_pos < 0 || a.L_pos > 10 || a.L_dir < 0 || a.L_dir > 4 || a.L_time < 0 || a.L_time > 10)
        return sw_fail(SWNERF_E_UNSUPP, "render_pass: embedder bands (%d,%d,%d) exceed (10,4,10)", a.L_pos, a.L_dir, a.L_time);
    if (a.z_vals && a.t_rand) return sw_fail(SWNERF_E_ARG, "render_pass: t_rand only applies to coarse sampling");
    PassDev P;
    P.a = a;
    int rc = noview ? stream_ptrs_noview(a.packed, a.out_ch, &P.w0, &P.b0, &P.nbias, &P.two_pass)
                    : stream_ptrs(a.kind, a.packed, a.run_deform, &P.w0, &P.b0, &P.nbias, &P.two_pass);
    if (rc) return rc;
    P.sort_n = 0; P.sort_s = 0;
    P.dir_steps = noview ? 0 : SW_STEPS_DIR;
    P.time_steps = (a.kind == SWNERF_NET_DNERF && P.two_pass) ? SW_STEPS_TIME : 0;
    P.tb_off = 0;
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
    if (P.time_steps) {                          // the four waves' per-ray TIME tiles, behind everything else
        P.tb_off = (int)(lds / sizeof(float));
        lds += 4 * SW_TB_LDS_FLOATS * sizeof(float);
    }
    const dim3 grid((unsigned)((a.n_rays + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    pass_startup_args(P, grid.x, noview ? SW_NOVIEW_STEPS : (a.kind == SWNERF_NET_DNERF && P.two_pass ? SW_DEFORM_STEPS + SW_CANON_STEPS : SW_CANON_STEPS));
    if (noview) hipLaunchKernelGGL((render_pass_kernel<false, false, 0, false>), grid, block, lds, st, P);
    else if (a.kind == SWNERF_NET_DNERF) hipLaunchKernelGGL(render_pass_kernel<true>, grid, block, lds, st, P);
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
    P.wvl = views_loop_ptr(kind, packed);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((M + 127) / 128)), block(256);
    const size_t lds = SW_LDS_FIXED_FLOATS * sizeof(float);
    if (kind == SWNERF_NET_DNERF) hipLaunchKernelGGL(mlp_forward_kernel<true>, grid, block, lds, st, P);
    else hipLaunchKernelGGL(mlp_forward_kernel<false>, grid, block, lds, st, P);
    return sw_check(hipGetLastError(), "mlp_forward launch");
}

// model.forward(x) for the net WITHOUT view directions (SWNERF_NET_NOVIEW; model.py:39-47, 59-60): x [M, C_pos] (any
// further columns of a wider row are ignored, like torch.split's second half with input_ch_views = 0), out [M, out_ch].
struct MlpNoviewDev { const float* x; int64_t M; int C; int Lp; const float* w0; const float* b0; int nbias; int out_ch; float* out; };

__global__ void __launch_bounds__(256, 1) mlp_forward_noview_kernel(MlpNoviewDev P) {
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
    float head[3];
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        const int col = sw_pos_col(a, h, P.Lp);
        emb[a >> 4][a & 15] = (col >= 0) ? xr[col] : 0.f;
    }
    WStream ws;
    ws_start(ws, P.w0, lds_bias, lds_ring, lane);
    trunk_pass<false, false, false, true>(emb, lds_emb, 0.f, false, h, in, out, head, ws);
    float o5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    head_valu_rt<8>(in, ws, P.out_ch, o5);
    if (live && h == 0) {
        float* o = P.out + row * P.out_ch;
        o[0] = o5[0] + ws.bias[0]; o[1] = o5[1] + ws.bias[1]; o[2] = o5[2] + ws.bias[2]; o[3] = o5[3] + ws.bias[3];
        if (P.out_ch == 5) o[4] = o5[4] + ws.bias[4];
    }
}

// vallina_NeRF.forward on embedded rows for use_viewdirs=False (model.py:39-47, 59-60): x [M, ldx >= C_pos] -> out [M, out_ch]
extern "C" int swnerf_mlp_forward_noview(const float* packed, const float* x, int64_t M, int ldx, int L_pos, int out_ch, float* out, void* stream) {
    if (M == 0 && packed) return 0;
    if (!packed || !x || !out || M < 0) return sw_fail(SWNERF_E_ARG, "mlp_forward_noview: NULL pointer or negative M");
    if (L_pos < 0 || L_pos > 10) return sw_fail(SWNERF_E_UNSUPP, "mlp_forward_noview: %d position bands exceed 10", L_pos);
    if (ldx < 3 * (1 + 2 * L_pos)) return sw_fail(SWNERF_E_ARG, "mlp_forward_noview: rows of %d floats cannot hold %d embedded columns", ldx, 3 * (1 + 2 * L_pos));
    MlpNoviewDev P;
    P.x = x; P.M = M; P.C = ldx; P.Lp = L_pos; P.out_ch = out_ch; P.out = out;
    int two = 0;
    int rc = stream_ptrs_noview(packed, out_ch, &P.w0, &P.b0, &P.nbias, &two);
    if (rc) return rc;
    hipLaunchKernelGGL(mlp_forward_noview_kernel, dim3((unsigned)((M + 127) / 128)), dim3(256), SW_LDS_FIXED_FLOATS * sizeof(float), (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "mlp_forward_noview launch");
}
