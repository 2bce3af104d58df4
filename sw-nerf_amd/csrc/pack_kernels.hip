// pack_kernels.hip - state_dict tensors -> the MFMA-fragment weight stream of mlp_core.h.
//
// For every GEMM segment (NT output tiles x KT input tiles of 32) the stream holds, per step
// (n, kt, q) and lane (i = lane&31, h = lane>>5), the float4
//     W[32n + i][ col(kt, r = 4q + e, h) ],  e = 0..3
// i.e. exactly the A operand of the e-th of four consecutive v_mfma_f32_32x32x2_f32.
// col() follows the B-operand slot of the activations: for a trunk k-tile register r of lane
// half h carries feature 32*kt + sw_frow(r,h) (it is the previous layer's accumulator);
// for an embedding k-tile the slot maps of swnerf_common.h apply.  Rows >= out_dim and pad
// slots get 0.  Biases are stored per output tile as [h][r] = b[32n + sw_frow(r,h)] so a lane
// reads the 16 initial accumulator values of its half with four float4 loads.
//
// Reference shapes: model.py:22-37 (vallina_NeRF), :251-269 (NeRFOriginal), :108-126 (_time net).
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "swnerf_common.h"
#include "host_util.h"

enum { KT_TRUNK = 0, KT_POS0 = 1, KT_POS1 = 2, KT_DIR = 3, KT_TIME = 4 };

struct PackSeg {
    const float* W; const float* b;     // b == NULL: accumulate segment, no bias tile
    int out_dim, in_dim, NT, KT;
    int ktype[10], kbase[10];
    int Lp, Ld, Lt;
    float* dstW; float* dstB;
    // transpose != 0 (backward stream): the GEMM's output rows are W's COLUMNS row0.. (out_dim of them) and
    // its k index runs over W's ROWS (kvalid of them): value = W[kbase + col][row0 + row]
    int transpose, row0, kvalid;
    // rowmap != 0 (transposed only): output row (n, i) is the position-embedding SLOT the lane holding it
    // filled in the forward pass - W column row0 + sw_pos_col(16n + r, h) with frow(r,h) == i - so the dX chain
    // leaves d gamma(x) in the registers pe_pos() wrote gamma(x) to.
    int rowmap;
};

__global__ void __launch_bounds__(256) pack_seg_kernel(PackSeg s) {
    const int nsteps = s.NT * s.KT * 4;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < nsteps * SW_STEP_FLOATS) {
        const int step = e / SW_STEP_FLOATS, rem = e % SW_STEP_FLOATS;
        const int lane = rem >> 2, i4 = rem & 3;
        const int n = step / (s.KT * 4), kt = (step >> 2) % s.KT, q = step & 3;
        const int r = 4 * q + i4, i = lane & 31, h = lane >> 5;
        const int row = 32 * n + i;
        int col = -1;
        switch (s.ktype[kt]) {
            case KT_TRUNK: col = sw_frow(r, h); break;
            case KT_POS0: col = sw_pos_col(r, h, s.Lp); break;
            case KT_POS1: col = sw_pos_col(16 + r, h, s.Lp); break;
            case KT_DIR: col = sw_dir_col(r, h, s.Ld); break;
            case KT_TIME: col = sw_time_col(r, h, s.Lt); break;
        }
        float v = 0.f;
        if (!s.transpose) {
            if (row < s.out_dim && col >= 0) v = s.W[(size_t)row * s.in_dim + s.kbase[kt] + col];
        } else {
            const int k = s.kbase[kt] + col;
            int wcol = row;
            if (s.rowmap) {
                const int hh = (i >> 2) & 1, rr = (i & 3) + 4 * (i >> 3);      // inverse of sw_frow
                wcol = sw_pos_col(16 * n + rr, hh, s.Lp);
            }
            if (wcol >= 0 && wcol < s.out_dim && k < s.kvalid) v = s.W[(size_t)k * s.in_dim + s.row0 + wcol];
        }
        s.dstW[e] = v;
    }
    if (s.b && e < s.NT * SW_BIAS_TILE_FLOATS) {
        const int n = e / SW_BIAS_TILE_FLOATS, rem = e % SW_BIAS_TILE_FLOATS;
        const int h = rem >> 4, r = rem & 15;
        const int row = 32 * n + sw_frow(r, h);
        s.dstB[e] = (row < s.out_dim) ? s.b[row] : 0.f;
    }
}

// head-bias tile: [h][r] = (b_a ++ b_b)[r] for r < n_a + n_b, the same in both lane halves, else 0
__global__ void pack_headbias_kernel(float* dst, const float* b_a, int n_a, const float* b_b, int n_b) {
    const int e = threadIdx.x;
    if (e >= SW_BIAS_TILE_FLOATS) return;
    const int r = e & 15;
    dst[e] = r < n_a ? b_a[r] : ((r - n_a) < n_b ? b_b[r - n_a] : 0.f);
}

// feature_linear folded into views_linears.0 (swnerf_common.h, SW_CANON_STEPS): fold[u][i] = sum_o Wv[u][o] W_f[o][i] for
// i < 256, fold[u][256 + c] = Wv[u][256 + c] for the Cdir view-direction columns, and behind the matrix
// b_vf[u] = sum_o Wv[u][o] b_f[o] + b_v[u].  Double accumulation, one rounding: the folded row is as close to the exact
// product as a float can be (the two-layer form rounds `feature` AND the second product).  Block u, thread i.
// Reference: model.py:49-53 (feature_linear -> cat -> views_linears[0], no activation in between).
__global__ void __launch_bounds__(256) fold_views_kernel(const float* Wv, int ldv, const float* bv, const float* Wf, const float* bf,
                                                         int Cdir, float* fold) {
    __shared__ float wrow[256];
    const int u = blockIdx.x, i = threadIdx.x;
    wrow[i] = Wv[(size_t)u * ldv + i];
    __syncthreads();
    double acc = 0.0;
    for (int o = 0; o < 256; ++o) acc += (double)wrow[o] * (double)Wf[(size_t)o * 256 + i];
    float* row = fold + (size_t)u * SW_FOLD_LD;
    row[i] = (float)acc;
    if (i < SW_FOLD_LD - 256) row[256 + i] = i < Cdir ? Wv[(size_t)u * ldv + 256 + i] : 0.f;
    if (i == 0) {
        double b = (double)bv[u];
        for (int o = 0; o < 256; ++o) b += (double)wrow[o] * (double)bf[o];
        fold[(size_t)128 * SW_FOLD_LD + u] = (float)b;
    }
}

// views_linears.0 . feature_linear of the net `params` (canonical order) -> fold [SW_FOLD_FLOATS]
static int fold_views(const float* const* params, int Cdir, float* fold, hipStream_t st) {
    hipLaunchKernelGGL(fold_views_kernel, dim3(128), dim3(256), 0, st, params[16], 256 + Cdir, params[17], params[18], params[19], Cdir, fold);
    return sw_check(hipGetLastError(), "pack_net fold launch");
}

struct Packer {
    hipStream_t st; float* w; float* b; int Lp, Ld, Lt; int rc;
    void seg(const float* W, const float* bias, int out_dim, int in_dim, int NT, int KT, const int* kt, const int* kb) {
        if (rc) return;
        PackSeg s;
        s.W = W; s.b = bias; s.out_dim = out_dim; s.in_dim = in_dim; s.NT = NT; s.KT = KT;
        for (int i = 0; i < 10; ++i) { s.ktype[i] = i < KT ? kt[i] : 0; s.kbase[i] = i < KT ? kb[i] : 0; }
        s.Lp = Lp; s.Ld = Ld; s.Lt = Lt; s.dstW = w; s.dstB = b;
        s.transpose = 0; s.row0 = 0; s.kvalid = 0; s.rowmap = 0;
        launch(s, NT, KT, bias != nullptr);
    }
    // transposed segment of the backward stream: out rows = W columns [row0, row0+n_out), k = W rows [0, kvalid)
    void segT(const float* W, int w_rows, int w_cols, int row0, int n_out, int NT, int KT, int rowmap = 0) {
        if (rc) return;
        PackSeg s;
        s.W = W; s.b = nullptr; s.out_dim = n_out; s.in_dim = w_cols; s.NT = NT; s.KT = KT;
        for (int i = 0; i < 10; ++i) { s.ktype[i] = KT_TRUNK; s.kbase[i] = 32 * i; }
        s.Lp = Lp; s.Ld = Ld; s.Lt = Lt; s.dstW = w; s.dstB = b;
        s.transpose = 1; s.row0 = row0; s.kvalid = w_rows; s.rowmap = rowmap;
        launch(s, NT, KT, false);
    }
    void launch(PackSeg& s, int NT, int KT, bool has_bias) {
        const int total = NT * KT * 4 * SW_STEP_FLOATS;
        hipLaunchKernelGGL(pack_seg_kernel, dim3((total + 255) / 256), dim3(256), 0, st, s);
        rc = sw_check(hipGetLastError(), "pack_net launch");
        w += total;
        if (has_bias) b += NT * SW_BIAS_TILE_FLOATS;
    }
    // `nout` weight rows of length `in_dim` (a multiple of 32) as bias-style tiles, for head_valu
    void vecs(const float* W, int nout, int in_dim) {
        for (int o = 0; o < nout && !rc; ++o) {
            PackSeg s;
            s.W = W; s.b = W + (size_t)o * in_dim; s.out_dim = in_dim; s.in_dim = in_dim; s.NT = in_dim / 32; s.KT = 0;
            for (int i = 0; i < 10; ++i) { s.ktype[i] = 0; s.kbase[i] = 0; }
            s.Lp = s.Ld = s.Lt = 0; s.dstW = w; s.dstB = b; s.transpose = 0; s.row0 = 0; s.kvalid = 0; s.rowmap = 0;
            hipLaunchKernelGGL(pack_seg_kernel, dim3((s.NT * SW_BIAS_TILE_FLOATS + 255) / 256), dim3(256), 0, st, s);
            rc = sw_check(hipGetLastError(), "pack_net launch");
            b += s.NT * SW_BIAS_TILE_FLOATS;
        }
    }
    // NT bias tiles of `bias` [out_dim] alone (no weight steps)
    void btiles(const float* bias, int out_dim, int NT) {
        if (rc) return;
        PackSeg s;
        s.W = bias; s.b = bias; s.out_dim = out_dim; s.in_dim = out_dim; s.NT = NT; s.KT = 0;
        for (int i = 0; i < 10; ++i) { s.ktype[i] = 0; s.kbase[i] = 0; }
        s.Lp = s.Ld = s.Lt = 0; s.dstW = w; s.dstB = b; s.transpose = 0; s.row0 = 0; s.kvalid = 0; s.rowmap = 0;
        hipLaunchKernelGGL(pack_seg_kernel, dim3((NT * SW_BIAS_TILE_FLOATS + 255) / 256), dim3(256), 0, st, s);
        rc = sw_check(hipGetLastError(), "pack_net launch");
        b += NT * SW_BIAS_TILE_FLOATS;
    }
    void headbias(const float* b_a, int n_a, const float* b_b, int n_b) {
        if (rc) return;
        hipLaunchKernelGGL(pack_headbias_kernel, dim3(1), dim3(64), 0, st, b, b_a, n_a, b_b, n_b);
        rc = sw_check(hipGetLastError(), "pack_net launch");
        b += SW_BIAS_TILE_FLOATS;
    }
    // one 8-layer trunk; P = {W0,b0,...,W7,b7}.  The head (Wh [head_out,256], bh) goes to the bias tiles.
    void trunk(const float* const* P, const float* Wh, const float* bh, int head_out, int Cpos, int Ctime,
               const float* bh2 = nullptr, int n2 = 0) {
        const int t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int b8[8];
        for (int i = 0; i < 8; ++i) b8[i] = 32 * i;
        const int e3[3] = {KT_POS0, KT_POS1, KT_TIME};
        const int eb[3] = {0, 0, Cpos};
        seg(P[0], P[1], 256, Cpos + Ctime, 8, Ctime ? 3 : 2, e3, eb);           // layer 0
        for (int l = 1; l < 8; ++l) {
            if (l == 5) {                                                          // input = cat[pts_emb, h]
                int b5[8];
                for (int i = 0; i < 8; ++i) b5[i] = Cpos + 32 * i;
                seg(P[10], P[11], 256, Cpos + 256, 8, 8, t8, b5);
                seg(P[10], nullptr, 256, Cpos + 256, 8, 2, e3, eb);
            } else {
                seg(P[2 * l], P[2 * l + 1], 256, 256, 8, 8, t8, b8);
            }
        }
        vecs(Wh, head_out, 256);
        headbias(bh, head_out, bh2, n2);
    }
};

extern "C" int swnerf_pack_net(int kind, const float* const* params, int L_pos, int L_dir, int L_time, float* packed, void* stream) {
    if (!params || !packed) return sw_fail(SWNERF_E_ARG, "pack_net: NULL pointer");
    if (kind != SWNERF_NET_CANON && kind != SWNERF_NET_DNERF) return sw_fail(SWNERF_E_ARG, "pack_net: unknown kind %d", kind);
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4 || L_time < 0 || L_time > 10)
        return sw_fail(SWNERF_E_UNSUPP, "pack_net: embedder bands (%d,%d,%d) exceed (10,4,10)", L_pos, L_dir, L_time);
    const int np = kind == SWNERF_NET_CANON ? 24 : 42;
    for (int i = 0; i < np; ++i) if (!params[i]) return sw_fail(SWNERF_E_ARG, "pack_net: params[%d] is NULL", i);
    const int Cpos = 3 * (1 + 2 * L_pos), Cdir = 3 * (1 + 2 * L_dir), Ctime = 1 + 2 * L_time;
    hipStream_t st = (hipStream_t)stream;
    int vt[9], vb[9];
    for (int i = 0; i < 8; ++i) { vt[i] = KT_TRUNK; vb[i] = 32 * i; }
    vt[8] = KT_DIR; vb[8] = 256;
    // feature_linear folded into views_linears.0: W_vf | Wv[:, 256:] and b_vf, kept at the end of the (last) CANON blob
    float* fold = packed + (kind == SWNERF_NET_DNERF ? SW_DNERF_A_FLOATS : 0) + SW_CANON_FOLD_OFFSET;
    int rc = fold_views(params, Cdir, fold, st);
    if (rc) return rc;
    auto canon = [&](Packer& pk) {
        pk.trunk(params, params[20], params[21], 1, Cpos, 0, params[23], 3);       // ... alpha_linear + head biases
        pk.seg(fold, fold + 128 * SW_FOLD_LD, 128, SW_FOLD_LD, 4, 9, vt, vb);      // VIEWSF on [h7 | gamma(d)]
        pk.vecs(params[22], 3, 128);                                               // rgb_linear.weight
    };
    auto tail = [&](float* wbase, const float* head) {
        return sw_check(hipMemcpyAsync(wbase, head, (size_t)SW_TAIL * SW_STEP_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net tail copy");
    };

    if (kind == SWNERF_NET_DNERF) {
        Packer pk{st, packed, packed + SW_DNERF_W_FLOATS, L_pos, L_dir, L_time, 0};
        pk.trunk(params + 24, params[40], params[41], 3, Cpos, Ctime);             // deformation net
        canon(pk);
        if (pk.rc) return pk.rc;
        if (pk.w != packed + (size_t)(SW_DEFORM_STEPS + SW_CANON_STEPS) * SW_STEP_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net: internal layout mismatch");
        if ((rc = tail(pk.w, packed))) return rc;
        packed += SW_DNERF_A_FLOATS;                                               // then the canon-only blob
    }
    Packer pk{st, packed, packed + SW_CANON_W_FLOATS, L_pos, L_dir, L_time, 0};
    canon(pk);
    if (pk.rc) return pk.rc;
    if (pk.w != packed + (size_t)SW_CANON_STEPS * SW_STEP_FLOATS || pk.b != packed + SW_CANON_VL_OFFSET)
        return sw_fail(SWNERF_E_ARG, "pack_net: internal layout mismatch");
    if ((rc = tail(pk.w, packed))) return rc;
    // the view branch once more, as a stream that wraps onto itself (biases: the tiles packed above)
    Packer vl{st, packed + SW_CANON_VL_OFFSET, nullptr, L_pos, L_dir, L_time, 0};
    vl.seg(fold, nullptr, 128, SW_FOLD_LD, 4, 9, vt, vb);
    if (vl.rc) return vl.rc;
    return tail(vl.w, packed + SW_CANON_VL_OFFSET);
}

// The canonical net's bias / head tiles in the UNFOLDED order (SW_X3_CANON_BIAS_TILES): what the bf16x3 core consumes, which
// still runs feature_linear as its own layer (mlp_core_x3.h).  Called by swnerf_pack_net_x3_kind (x3_kernels.hip).
int sw_pack_canon_bias_unfolded(const float* const* params, float* dst, hipStream_t st) {
    Packer pk{st, nullptr, dst, 0, 0, 0, 0};
    for (int l = 0; l < 8; ++l) pk.btiles(params[2 * l + 1], 256, 8);                  // pts_linears.l.bias
    pk.vecs(params[20], 1, 256);                                                       // alpha_linear.weight
    pk.headbias(params[21], 1, params[23], 3);                                         // [b_alpha, b_r, b_g, b_b]
    pk.btiles(params[19], 256, 8);                                                     // feature_linear.bias
    pk.btiles(params[17], 128, 4);                                                     // views_linears.0.bias
    pk.vecs(params[22], 3, 128);                                                       // rgb_linear.weight
    if (!pk.rc && pk.b != dst + SW_X3_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net_x3: bias layout mismatch");
    return pk.rc;
}

// use_viewdirs=False (model.py:59-60): the trunk alone; output_linear's rows ride as bias-style tiles like the other heads.
extern "C" int swnerf_pack_net_noview(const float* const* params, int L_pos, int out_ch, float* packed, void* stream) {
    if (!params || !packed) return sw_fail(SWNERF_E_ARG, "pack_net_noview: NULL pointer");
    if (L_pos < 0 || L_pos > 10) return sw_fail(SWNERF_E_UNSUPP, "pack_net_noview: %d position bands exceed 10", L_pos);
    if (out_ch < 4 || out_ch > SW_NOVIEW_MAX_OUT)
        return sw_fail(SWNERF_E_UNSUPP, "pack_net_noview: output_ch %d (the reference builds 4 or 5, nerf/run.py:231)", out_ch);
    for (int i = 0; i < 18; ++i) if (!params[i]) return sw_fail(SWNERF_E_ARG, "pack_net_noview: params[%d] is NULL", i);
    hipStream_t st = (hipStream_t)stream;
    Packer pk{st, packed, packed + SW_NOVIEW_W_FLOATS, L_pos, 0, 0, 0};
    pk.trunk(params, params[16], params[17], out_ch, 3 * (1 + 2 * L_pos), 0);
    if (pk.rc) return pk.rc;
    if (pk.w != packed + (size_t)SW_NOVIEW_STEPS * SW_STEP_FLOATS || pk.b != packed + SW_NOVIEW_W_FLOATS + SW_NOVIEW_BIAS_TILES(out_ch) * SW_BIAS_TILE_FLOATS)
        return sw_fail(SWNERF_E_ARG, "pack_net_noview: internal layout mismatch");
    return sw_check(hipMemcpyAsync(pk.w, packed, (size_t)SW_TAIL * SW_STEP_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net_noview tail copy");
}

// The backward (dX chain) streams: transposed weights in the order the backward kernels consume them
// (swnerf_common.h SW_BWD_* / SW_DBWD_*), then the head weights as bias-style tiles.
extern "C" size_t swnerf_packed_bwd_floats_kind(int bwd_kind) {
    switch (bwd_kind) {
        case SWNERF_BWD_CANON: return (size_t)SW_BWD_FLOATS;
        case SWNERF_BWD_CANON_INPUT_GRAD: return (size_t)SW_BWD_IG_FLOATS;
        case SWNERF_BWD_DEFORM: return (size_t)SW_DBWD_FLOATS;
        case SWNERF_BWD_DNERF_FUSED: return (size_t)SW_BWD_DN_FLOATS;
    }
    return 0;
}

extern "C" int swnerf_pack_net_bwd_kind(int bwd_kind, const float* const* params, int L_pos, int L_dir, float* packed_bwd, void* stream) {
    if (!params || !packed_bwd) return sw_fail(SWNERF_E_ARG, "pack_net_bwd: NULL pointer");
    if (bwd_kind != SWNERF_BWD_CANON && bwd_kind != SWNERF_BWD_CANON_INPUT_GRAD && bwd_kind != SWNERF_BWD_DEFORM && bwd_kind != SWNERF_BWD_DNERF_FUSED)
        return sw_fail(SWNERF_E_ARG, "pack_net_bwd: unknown stream kind %d", bwd_kind);
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "pack_net_bwd: embedder bands (%d,%d) exceed (10,4)", L_pos, L_dir);
    const int np = bwd_kind == SWNERF_BWD_DEFORM ? 18 : (bwd_kind == SWNERF_BWD_DNERF_FUSED ? 42 : 24);
    for (int i = 0; i < np; ++i) if (!params[i]) return sw_fail(SWNERF_E_ARG, "pack_net_bwd: params[%d] is NULL", i);
    const int Cpos = 3 * (1 + 2 * L_pos), Cdir = 3 * (1 + 2 * L_dir);
    hipStream_t st = (hipStream_t)stream;
    auto tail = [&](Packer& pk, size_t steps) {
        if (pk.rc) return pk.rc;
        if (pk.w != packed_bwd + steps * SW_STEP_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net_bwd: internal layout mismatch");
        return sw_check(hipMemcpyAsync(pk.w, packed_bwd, (size_t)SW_TAIL * SW_STEP_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net_bwd tail copy");
    };
    if (bwd_kind == SWNERF_BWD_DEFORM) {
        Packer pk{st, packed_bwd, packed_bwd + SW_DBWD_W_FLOATS, L_pos, L_dir, 0, 0};
        for (int l = 7; l >= 1; --l)                                  // _time.l.weight[:, -256:]^T
            pk.segT(params[2 * l], 256, l == 5 ? Cpos + 256 : 256, l == 5 ? Cpos : 0, 256, 8, 8);
        int rc = tail(pk, SW_DBWD_STEPS);
        if (rc) return rc;
        pk.vecs(params[16], 3, 256);                                  // _time_out.weight rows, tile n: [h][r] = w[o][32n + frow(r,h)]
        return pk.rc;
    }
    const bool fused = bwd_kind == SWNERF_BWD_DNERF_FUSED;
    const bool ig = bwd_kind == SWNERF_BWD_CANON_INPUT_GRAD || fused;
    const size_t wfloats = fused ? SW_BWD_DN_W_FLOATS : (ig ? SW_BWD_IG_W_FLOATS : SW_BWD_W_FLOATS);
    Packer pk{st, packed_bwd, packed_bwd + wfloats, L_pos, L_dir, 0, 0};
    // d h7 = W_vf^T . d pre_hv (the fold of swnerf_pack_net, recomputed into this blob's scratch behind the bias tiles)
    float* fold = packed_bwd + wfloats + (SW_BWD_BIAS_TILES + (fused ? 24 : 0)) * SW_BIAS_TILE_FLOATS;
    int rcf = fold_views(params, Cdir, fold, st);
    if (rcf) return rcf;
    pk.segT(params[22], 3, 128, 0, 128, 4, 1);                       // rgb_linear.weight [3,128]^T
    pk.segT(fold, 128, SW_FOLD_LD, 0, 256, 8, 4);                    // W_vf^T  (views_linears.0[:, :256] . feature_linear)
    for (int l = 7; l >= 1; --l) {                                   // pts_linears.l.weight[:, -256:]^T
        if (ig && l == 5) pk.segT(params[10], 256, Cpos + 256, 0, Cpos, 2, 8, 1);   // ... [:, :Cpos]^T -> d gamma(x)
        pk.segT(params[2 * l], 256, l == 5 ? Cpos + 256 : 256, l == 5 ? Cpos : 0, 256, 8, 8);
    }
    if (ig) pk.segT(params[0], 256, Cpos, 0, Cpos, 2, 8, 1);         // pts_linears.0.weight^T -> d gamma(x)
    if (fused)                                                       // ... then the deformation net's chain in the same ring
        for (int l = 7; l >= 1; --l)
            pk.segT(params[24 + 2 * l], 256, l == 5 ? Cpos + 256 : 256, l == 5 ? Cpos : 0, 256, 8, 8);
    int rc = tail(pk, fused ? SW_BWD_DN_STEPS : (ig ? SW_BWD_IG_STEPS : SW_BWD_STEPS));
    if (rc) return rc;
    pk.vecs(params[20], 1, 256);                                     // alpha_linear.weight [1,256] as 8 bias-style tiles
    if (fused) pk.vecs(params[40], 3, 256);                          // _time_out.weight rows as 3 x 8 tiles
    return pk.rc;
}

// The dX chain's stream of the net without view directions: pts_linears.7 .. .1 transposed (trunk columns), then
// output_linear.weight rows as bias-style tiles.  params as swnerf_pack_net_noview.
extern "C" size_t swnerf_packed_bwd_noview_floats(void) { return (size_t)SW_NVBWD_FLOATS; }
extern "C" int swnerf_pack_net_bwd_noview(const float* const* params, int L_pos, int out_ch, float* packed_bwd, void* stream) {
    if (!params || !packed_bwd) return sw_fail(SWNERF_E_ARG, "pack_net_bwd_noview: NULL pointer");
    if (L_pos < 0 || L_pos > 10) return sw_fail(SWNERF_E_UNSUPP, "pack_net_bwd_noview: %d position bands exceed 10", L_pos);
    if (out_ch < 4 || out_ch > SW_NOVIEW_MAX_OUT) return sw_fail(SWNERF_E_UNSUPP, "pack_net_bwd_noview: output_ch %d (4 or 5)", out_ch);
    for (int i = 0; i < 18; ++i) if (!params[i]) return sw_fail(SWNERF_E_ARG, "pack_net_bwd_noview: params[%d] is NULL", i);
    const int Cpos = 3 * (1 + 2 * L_pos);
    hipStream_t st = (hipStream_t)stream;
    Packer pk{st, packed_bwd, packed_bwd + SW_DBWD_W_FLOATS, L_pos, 0, 0, 0};
    for (int l = 7; l >= 1; --l)                                  // pts_linears.l.weight[:, -256:]^T
        pk.segT(params[2 * l], 256, l == 5 ? Cpos + 256 : 256, l == 5 ? Cpos : 0, 256, 8, 8);
    if (pk.rc) return pk.rc;
    if (pk.w != packed_bwd + (size_t)SW_DBWD_STEPS * SW_STEP_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net_bwd_noview: internal layout mismatch");
    int rc = sw_check(hipMemcpyAsync(pk.w, packed_bwd, (size_t)SW_TAIL * SW_STEP_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net_bwd_noview tail copy");
    if (rc) return rc;
    pk.vecs(params[16], out_ch, 256);                             // output_linear.weight rows, tile n: [h][r] = w[c][32n + frow(r,h)]
    return pk.rc;
}

extern "C" int swnerf_pack_net_bwd(const float* const* params, int L_pos, int L_dir, float* packed_bwd, void* stream) {
    return swnerf_pack_net_bwd_kind(SWNERF_BWD_CANON, params, L_pos, L_dir, packed_bwd, stream);
}
