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

enum { XK_TRUNK = 0, XK_POS0 = 1, XK_POS1 = 2, XK_DIR = 3, XK_TIME = 4 };

int sw_pack_canon_bias_x3(const float* const* params, const float* fold, float* dst, hipStream_t st);   // pack_kernels.hip

struct X3Seg {
    const float* W; int out_dim, in_dim, NT, KB;
    int ktype[20], kbase[20];       // per k-block: slot map and the first weight column of its tile
    int Lp, Ld, Lt;
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
            case XK_TIME: col = sw_time_col(r, h, s.Lt); break;
        }
        float v = 0.f;
        if (row < s.out_dim && col >= 0) v = s.W[(size_t)row * s.in_dim + s.kbase[kb] + col];
        const unsigned hi = x3_bf16_rne(v);
        const unsigned b = which ? x3_bf16_rne(v - __uint_as_float(hi << 16)) : hi;
        out |= b << (16 * t);
    }
    s.dst[e] = out;
}

extern "C" size_t swnerf_packed_x3_floats_kind(int kind) {
    return kind == SWNERF_NET_CANON ? (size_t)SW_X3_FLOATS : (kind == SWNERF_NET_DNERF ? (size_t)SW_X3_DNERF_FLOATS : 0);
}
extern "C" size_t swnerf_packed_x3_floats(void) { return (size_t)SW_X3_FLOATS; }

struct X3Packer {
    hipStream_t st; unsigned* w; int Lp, Ld, Lt, Cpos, Cdir, Ctime; int rc;
    int kt[20], kb0[20];
    void seg(const float* W, int out_dim, int in_dim, int NT, int KB) {
        if (rc) return;
        X3Seg s;
        s.W = W; s.out_dim = out_dim; s.in_dim = in_dim; s.NT = NT; s.KB = KB; s.Lp = Lp; s.Ld = Ld; s.Lt = Lt; s.dst = w;
        for (int i = 0; i < 20; ++i) { s.ktype[i] = i < KB ? kt[i] : 0; s.kbase[i] = i < KB ? kb0[i] : 0; }
        const int total = NT * KB * 512;
        hipLaunchKernelGGL(x3_pack_seg_kernel, dim3((total + 255) / 256), dim3(256), 0, st, s);
        rc = sw_check(hipGetLastError(), "pack_net_x3 launch");
        w += total;
    }
    void trunk_blocks(int base) { for (int i = 0; i < 16; ++i) { kt[i] = XK_TRUNK; kb0[i] = base + 32 * (i >> 1); } }
    void pos_blocks(int at) { for (int i = 0; i < 4; ++i) { kt[at + i] = i < 2 ? XK_POS0 : XK_POS1; kb0[at + i] = 0; } }
    // one 8-layer trunk; P = {W0,b0,...,W7,b7}; time != 0: layer 0 also takes gamma(t) (the deformation net)
    void trunk(const float* const* P, bool time) {
        pos_blocks(0);
        if (time) { kt[4] = kt[5] = XK_TIME; kb0[4] = kb0[5] = Cpos; }
        seg(P[0], 256, Cpos + (time ? Ctime : 0), 8, time ? 6 : 4);
        for (int l = 1; l < 8; ++l) {
            if (l == 5) {                                                           // input = cat[gamma(x), h]
                trunk_blocks(Cpos); pos_blocks(16);
                seg(P[10], 256, Cpos + 256, 8, 20);
            } else {
                trunk_blocks(0);
                seg(P[2 * l], 256, 256, 8, 16);
            }
        }
    }
    // fold: the fp32 blob's folded view layer [128][SW_FOLD_LD] = [Wv[:, :256] . W_f | Wv[:, 256:]] (feature_linear has no activation)
    void canon(const float* const* params, const float* fold) {
        trunk(params, false);
        trunk_blocks(0);
        kt[16] = kt[17] = XK_DIR; kb0[16] = kb0[17] = 256;
        seg(fold, 128, SW_FOLD_LD, 4, 18);                                          // views_linears.0 . feature_linear on [h7 | gamma(d)]
    }
};

// packed_fp32: the blob swnerf_pack_net made for the same kind and tensors (its deformation bias tiles and its folded view layer are used)
extern "C" int swnerf_pack_net_x3_kind(int kind, const float* const* params, int L_pos, int L_dir, int L_time,
                                       const float* packed_fp32, float* packed_x3, void* stream) {
    if (!params || !packed_fp32 || !packed_x3) return sw_fail(SWNERF_E_ARG, "pack_net_x3: NULL pointer");
    if (kind != SWNERF_NET_CANON && kind != SWNERF_NET_DNERF) return sw_fail(SWNERF_E_ARG, "pack_net_x3: unknown kind %d", kind);
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4 || L_time < 0 || L_time > 10)
        return sw_fail(SWNERF_E_UNSUPP, "pack_net_x3: embedder bands (%d,%d,%d) exceed (10,4,10)", L_pos, L_dir, L_time);
    const int np = kind == SWNERF_NET_CANON ? 24 : 42;
    for (int i = 0; i < np; ++i) if (!params[i]) return sw_fail(SWNERF_E_ARG, "pack_net_x3: params[%d] is NULL", i);
    hipStream_t st = (hipStream_t)stream;
    auto copy = [&](float* dst, const float* src, size_t floats, const char* what) {
        return sw_check(hipMemcpyAsync(dst, src, floats * sizeof(float), hipMemcpyDeviceToDevice, st), what);
    };
    int rc = 0;
    if (kind == SWNERF_NET_DNERF) {
        X3Packer pk{st, reinterpret_cast<unsigned*>(packed_x3), L_pos, L_dir, L_time, 3 * (1 + 2 * L_pos), 3 * (1 + 2 * L_dir), 1 + 2 * L_time, 0, {0}, {0}};
        pk.trunk(params + 24, true);                                                // deformation net, then the canonical net
        pk.canon(params, packed_fp32 + SW_DNERF_A_FLOATS + SW_CANON_FOLD_OFFSET);
        if (pk.rc) return pk.rc;
        if (pk.w != reinterpret_cast<unsigned*>(packed_x3) + (size_t)(SW_X3_DEFORM_CHUNKS + SW_X3_CANON_CHUNKS) * SW_X3_CHUNK_FLOATS)
            return sw_fail(SWNERF_E_ARG, "pack_net_x3: internal layout mismatch");
        if ((rc = copy(reinterpret_cast<float*>(pk.w), packed_x3, (size_t)SW_X3_TAIL_CHUNKS * SW_X3_CHUNK_FLOATS, "pack_net_x3 tail copy"))) return rc;
        // bias tiles: the deformation net's as in the fp32 blob, the canonical net's in this core's order (b_vf behind the head-bias tile)
        // (the fp32 blob's bias stream starts with the 4 b_vf tiles of its per-ray DIR prefix; the deformation net's follow)
        if ((rc = copy(packed_x3 + SW_X3_DNERF_W_FLOATS, packed_fp32 + SW_DNERF_W_FLOATS + SW_DIR_BIAS_TILES * SW_BIAS_TILE_FLOATS,
                       (size_t)SW_DEFORM_BIAS_TILES * SW_BIAS_TILE_FLOATS, "pack_net_x3 bias copy"))) return rc;
        if ((rc = sw_pack_canon_bias_x3(params, packed_fp32 + SW_DNERF_A_FLOATS + SW_CANON_FOLD_OFFSET,
                                        packed_x3 + SW_X3_DNERF_W_FLOATS + SW_DEFORM_BIAS_TILES * SW_BIAS_TILE_FLOATS, st))) return rc;
        packed_x3 += SW_X3_DNERF_A_FLOATS;                                          // then the canon-only blob (t == 0 branch)
        packed_fp32 += SW_DNERF_A_FLOATS;
    }
    X3Packer pk{st, reinterpret_cast<unsigned*>(packed_x3), L_pos, L_dir, L_time, 3 * (1 + 2 * L_pos), 3 * (1 + 2 * L_dir), 1 + 2 * L_time, 0, {0}, {0}};
    pk.canon(params, packed_fp32 + SW_CANON_FOLD_OFFSET);
    if (pk.rc) return pk.rc;
    if (pk.w != reinterpret_cast<unsigned*>(packed_x3) + (size_t)SW_X3_CANON_CHUNKS * SW_X3_CHUNK_FLOATS)
        return sw_fail(SWNERF_E_ARG, "pack_net_x3: internal layout mismatch");
    if ((rc = copy(reinterpret_cast<float*>(pk.w), packed_x3, (size_t)SW_X3_TAIL_CHUNKS * SW_X3_CHUNK_FLOATS, "pack_net_x3 tail copy"))) return rc;
    return sw_pack_canon_bias_x3(params, packed_fp32 + SW_CANON_FOLD_OFFSET, packed_x3 + SW_X3_W_FLOATS, st);
}

extern "C" int swnerf_pack_net_x3(const float* const* params, int L_pos, int L_dir, const float* packed_canon, float* packed_x3, void* stream) {
    return swnerf_pack_net_x3_kind(SWNERF_NET_CANON, params, L_pos, L_dir, 0, packed_canon, packed_x3, stream);
}

extern "C" int swnerf_render_pass_x3(const swnerf_pass_args* args, int terms, void* stream) {
    if (!args) return sw_fail(SWNERF_E_ARG, "render_pass_x3: NULL args");
    const swnerf_pass_args& a = *args;
    if (terms != 1 && terms != 3) return sw_fail(SWNERF_E_ARG, "render_pass_x3: terms must be 3 (bf16x3) or 1 (plain bf16), got %d", terms);
    if (!a.packed || (!a.ray_batch && a.n_rays != 0)) return sw_fail(SWNERF_E_ARG, "render_pass_x3: NULL ray_batch/packed");
    if (a.kind != SWNERF_NET_CANON && a.kind != SWNERF_NET_DNERF) return sw_fail(SWNERF_E_ARG, "render_pass_x3: unknown net kind %d", a.kind);
    if (a.n_rays < 0 || a.n_samples < 2) return sw_fail(SWNERF_E_ARG, "render_pass_x3: n_rays %lld, n_samples %d", (long long)a.n_rays, a.n_samples);
    if (a.cols != 11 && a.cols != 12) return sw_fail(SWNERF_E_ARG, "render_pass_x3: ray_batch must have 11 or 12 columns, got %d", a.cols);
    if (a.kind == SWNERF_NET_DNERF && a.cols != 12) return sw_fail(SWNERF_E_ARG, "render_pass_x3: D-NeRF needs the frame_time column");
    if (a.L_pos < 0 || a.L_pos > 10 || a.L_dir < 0 || a.L_dir > 4 || a.L_time < 0 || a.L_time > 10)
        return sw_fail(SWNERF_E_UNSUPP, "render_pass_x3: embedder bands (%d,%d,%d) exceed (10,4,10)", a.L_pos, a.L_dir, a.L_time);
    if (a.z_vals && a.t_rand) return sw_fail(SWNERF_E_ARG, "render_pass_x3: t_rand only applies to coarse sampling");
    const bool dn = a.kind == SWNERF_NET_DNERF;
    PassDev P = {};
    P.a = a;
    P.nbias = SW_X3_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS;
    if (!dn) { P.w0 = a.packed; P.b0 = a.packed + SW_X3_W_FLOATS; }
    else if (a.run_deform) { P.w0 = a.packed; P.b0 = a.packed + SW_X3_DNERF_W_FLOATS; P.two_pass = 1; P.nbias = (SW_DEFORM_BIAS_TILES + SW_X3_CANON_BIAS_TILES) * SW_BIAS_TILE_FLOATS; }
    else { const float* c = a.packed + SW_X3_DNERF_A_FLOATS; P.w0 = c; P.b0 = c + SW_X3_W_FLOATS; }
    size_t lds = (size_t)(dn ? X3Lds<true>::FIXED : X3Lds<false>::FIXED) * sizeof(float);
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
    if (!dn && a.dx) {
        int rc = sw_check(hipMemsetAsync(a.dx, 0, (size_t)a.n_rays * a.n_samples * 3 * sizeof(float), st), "render_pass_x3 dx fill");
        if (rc) return rc;
    }
    const dim3 grid((unsigned)((a.n_rays + 3) / 4)), block(256);
    if (dn) {
        if (terms == 3) hipLaunchKernelGGL((render_pass_kernel<true, false, 3>), grid, block, lds, st, P);
        else hipLaunchKernelGGL((render_pass_kernel<true, false, 1>), grid, block, lds, st, P);
    } else {
        if (terms == 3) hipLaunchKernelGGL((render_pass_kernel<false, false, 3>), grid, block, lds, st, P);
        else hipLaunchKernelGGL((render_pass_kernel<false, false, 1>), grid, block, lds, st, P);
    }
    return sw_check(hipGetLastError(), "render_pass_x3 launch");
}
