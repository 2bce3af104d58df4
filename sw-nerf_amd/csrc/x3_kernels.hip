// x3_kernels.hip - the opt-in bf16x3 render pass (mlp_core_x3.h): weights and activations split into two bf16 halves,
// three bf16 MFMAs per product term group, fp32 accumulation.  Static canonical net (vallina_NeRF / NeRFOriginal).
//
//   swnerf_pack_net_x3   : state_dict tensors -> the k-block-major [A_hi][A_lo] weight stream
//   swnerf_render_pass_x3: swnerf_render_pass with the MLP on the bf16 matrix pipe (same sampling, encoding,
//                          compositing and resampling code: render_pass.h, PREC != 0)
//
// A-operand fragment of group (kb, n): lane (i, h) holds W[32n + i][col(kb, 8c.. )] for the 8 k-slots p = 0..7 of its
// half - slot p of k-block kb = 2t + c is register r = 8c + p of B-operand tile t: feature 32t + sw_frow(r, h) for a
// hidden activation, the embedding slot maps of swnerf_common.h for gamma(x) / gamma(d).
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "swnerf_common.h"
#include "host_util.h"
#include "mlp_kernels.h"
#include "render_pass.h"

enum { XK_TRUNK = 0, XK_POS0 = 1, XK_POS1 = 2, XK_DIR = 3 };

struct X3Seg {
    const float* W; int out_dim, in_dim, NT, KB;
    int ktype[20], kbase[20];       // per k-block: slot map and the first weight column of its tile
    int Lp, Ld;
    unsigned* dst;
};

__device__ __forceinline__ unsigned x3_bf16_rne(float v) {            // finite inputs (weights)
    const unsigned u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__global__ void __launch_bounds__(256) x3_pack_seg_kernel(X3Seg s) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= s.NT * s.KB * 512) return;
    const int G = e >> 9, rem = e & 511, which = rem >> 8, lane = (rem & 255) >> 2, q = rem & 3;
    const int kb = G / s.NT, n = G % s.NT, i = lane & 31, h = lane >> 5, c = kb & 1;
    const int row = 32 * n + i;
    unsigned out = 0;
    for (int t = 0; t < 2; ++t) {
        const int r = 8 * c + 2 * q + t;
        int col = -1;
        switch (s.ktype[kb]) {
            case XK_TRUNK: col = sw_frow(r, h); break;
            case XK_POS0: col = sw_pos_col(r, h, s.Lp); break;
            case XK_POS1: col = sw_pos_col(16 + r, h, s.Lp); break;
            case XK_DIR: col = sw_dir_col(r, h, s.Ld); break;
        }
        float v = 0.f;
        if (row < s.out_dim && col >= 0) v = s.W[(size_t)row * s.in_dim + s.kbase[kb] + col];
        const unsigned hi = x3_bf16_rne(v);
        const unsigned b = which ? x3_bf16_rne(v - __uint_as_float(hi << 16)) : hi;
        out |= b << (16 * t);
    }
    s.dst[e] = out;
}

extern "C" size_t swnerf_packed_x3_floats(void) { return (size_t)SW_X3_FLOATS; }

extern "C" int swnerf_pack_net_x3(const float* const* params, int L_pos, int L_dir, const float* packed_canon, float* packed_x3, void* stream) {
    if (!params || !packed_canon || !packed_x3) return sw_fail(SWNERF_E_ARG, "pack_net_x3: NULL pointer");
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "pack_net_x3: embedder bands (%d,%d) exceed (10,4)", L_pos, L_dir);
    for (int i = 0; i < 24; ++i) if (!params[i]) return sw_fail(SWNERF_E_ARG, "pack_net_x3: params[%d] is NULL", i);
    const int Cpos = 3 * (1 + 2 * L_pos), Cdir = 3 * (1 + 2 * L_dir);
    hipStream_t st = (hipStream_t)stream;
    unsigned* w = reinterpret_cast<unsigned*>(packed_x3);
    int rc = 0;
    auto seg = [&](const float* W, int out_dim, int in_dim, int NT, int KB, const int* kt, const int* kbase) {
        if (rc) return;
        X3Seg s;
        s.W = W; s.out_dim = out_dim; s.in_dim = in_dim; s.NT = NT; s.KB = KB; s.Lp = L_pos; s.Ld = L_dir; s.dst = w;
        for (int i = 0; i < 20; ++i) { s.ktype[i] = i < KB ? kt[i] : 0; s.kbase[i] = i < KB ? kbase[i] : 0; }
        const int total = NT * KB * 512;
        hipLaunchKernelGGL(x3_pack_seg_kernel, dim3((total + 255) / 256), dim3(256), 0, st, s);
        rc = sw_check(hipGetLastError(), "pack_net_x3 launch");
        w += total;
    };
    int kt[20], kb0[20];
    auto trunk_blocks = [&](int base) { for (int i = 0; i < 16; ++i) { kt[i] = XK_TRUNK; kb0[i] = base + 32 * (i >> 1); } };
    auto pos_blocks = [&](int at) { for (int i = 0; i < 4; ++i) { kt[at + i] = i < 2 ? XK_POS0 : XK_POS1; kb0[at + i] = 0; } };
    pos_blocks(0);
    seg(params[0], 256, Cpos, 8, 4, kt, kb0);                                       // pts_linears.0
    for (int l = 1; l < 8; ++l) {
        if (l == 5) {                                                               // input = cat[gamma(x), h]
            trunk_blocks(Cpos); pos_blocks(16);
            seg(params[10], 256, Cpos + 256, 8, 20, kt, kb0);
        } else {
            trunk_blocks(0);
            seg(params[2 * l], 256, 256, 8, 16, kt, kb0);
        }
    }
    trunk_blocks(0);
    seg(params[18], 256, 256, 8, 16, kt, kb0);                                      // feature_linear
    kt[16] = kt[17] = XK_DIR; kb0[16] = kb0[17] = 256;
    seg(params[16], 128, 256 + Cdir, 4, 18, kt, kb0);                               // views_linears.0 = [feature | gamma(d)]
    if (rc) return rc;
    if (w != reinterpret_cast<unsigned*>(packed_x3) + (size_t)SW_X3_CANON_CHUNKS * SW_X3_CHUNK_FLOATS)
        return sw_fail(SWNERF_E_ARG, "pack_net_x3: internal layout mismatch");
    rc = sw_check(hipMemcpyAsync(w, packed_x3, (size_t)SW_X3_TAIL_CHUNKS * SW_X3_CHUNK_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net_x3 tail copy");
    if (rc) return rc;
    return sw_check(hipMemcpyAsync(packed_x3 + SW_X3_W_FLOATS, packed_canon + SW_CANON_W_FLOATS,
                                   (size_t)SW_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net_x3 bias copy");
}

// LDS of the x3 pass: bias tiles | shared weight ring | per wave: gamma(d) tile + depth slots | per wave: resampling scratch
#define X3_LDS_BIAS_FLOATS (SW_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS)

extern "C" int swnerf_render_pass_x3(const swnerf_pass_args* args, int terms, void* stream) {
    if (!args) return sw_fail(SWNERF_E_ARG, "render_pass_x3: NULL args");
    const swnerf_pass_args& a = *args;
    if (terms != 1 && terms != 3) return sw_fail(SWNERF_E_ARG, "render_pass_x3: terms must be 3 (bf16x3) or 1 (plain bf16), got %d", terms);
    if (!a.packed || (!a.ray_batch && a.n_rays != 0)) return sw_fail(SWNERF_E_ARG, "render_pass_x3: NULL ray_batch/packed");
    if (a.kind != SWNERF_NET_CANON) return sw_fail(SWNERF_E_UNSUPP, "render_pass_x3: static canonical net only (kind %d)", a.kind);
    if (a.n_rays < 0 || a.n_samples < 2) return sw_fail(SWNERF_E_ARG, "render_pass_x3: n_rays %lld, n_samples %d", (long long)a.n_rays, a.n_samples);
    if (a.cols != 11 && a.cols != 12) return sw_fail(SWNERF_E_ARG, "render_pass_x3: ray_batch must have 11 or 12 columns, got %d", a.cols);
    if (a.L_pos < 0 || a.L_pos > 10 || a.L_dir < 0 || a.L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "render_pass_x3: embedder bands (%d,%d) exceed (10,4)", a.L_pos, a.L_dir);
    if (a.z_vals && a.t_rand) return sw_fail(SWNERF_E_ARG, "render_pass_x3: t_rand only applies to coarse sampling");
    PassDev P = {};
    P.a = a;
    P.w0 = a.packed; P.b0 = a.packed + SW_X3_W_FLOATS; P.nbias = X3_LDS_BIAS_FLOATS; P.two_pass = 0;
    size_t lds = (size_t)(X3_LDS_BIAS_FLOATS + X3_RING_FLOATS + 4 * X3_WAVE_FLOATS) * sizeof(float);
    if (a.n_importance > 0) {
        if (!a.z_fine && a.n_rays != 0) return sw_fail(SWNERF_E_ARG, "render_pass_x3: n_importance>0 needs z_fine");
        if (a.n_samples < 3 || a.n_samples > SW_LDS_SC || a.n_samples + a.n_importance > SW_LDS_SORT)
            return sw_fail(SWNERF_E_UNSUPP, "render_pass_x3: resampling supports 3<=N_samples<=%d and N_samples+N_importance<=%d", SW_LDS_SC, SW_LDS_SORT);
        int p2 = 2;
        while (p2 < a.n_importance) p2 <<= 1;
        P.sort_n = p2;
        p2 = 2;
        while (p2 < a.n_samples) p2 <<= 1;
        P.sort_s = p2;
        lds += 4 * SW_LDS_WAVE_FLOATS * sizeof(float);
    }
    if (a.n_rays == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (a.dx) {
        int rc = sw_check(hipMemsetAsync(a.dx, 0, (size_t)a.n_rays * a.n_samples * 3 * sizeof(float), st), "render_pass_x3 dx fill");
        if (rc) return rc;
    }
    const dim3 grid((unsigned)((a.n_rays + 3) / 4)), block(256);
    if (terms == 3) hipLaunchKernelGGL((render_pass_kernel<false, false, 3>), grid, block, lds, st, P);
    else hipLaunchKernelGGL((render_pass_kernel<false, false, 1>), grid, block, lds, st, P);
    return sw_check(hipGetLastError(), "render_pass_x3 launch");
}
