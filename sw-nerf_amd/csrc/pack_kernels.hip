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

// One segment of a pack plan (kept small: a plan of up to PACK_MAX_SEGS of them travels as a kernel argument).
struct PackSeg {
    const float* W; const float* b;     // b == NULL: accumulate segment, no bias tile
    float* dstW; float* dstB;
    int out_dim, in_dim;
    short NT, KT;
    // transpose != 0 (backward stream): the GEMM's output rows are W's COLUMNS row0.. (out_dim of them) and
    // its k index runs over W's ROWS (kvalid of them): value = W[kbase + col][row0 + row]
    // rowmap != 0 (transposed only): output row (n, i) is the position-embedding SLOT the lane holding it
    // filled in the forward pass - W column row0 + sw_pos_col(16n + r, h) with frow(r,h) == i - so the dX chain
    // leaves d gamma(x) in the registers pe_pos() wrote gamma(x) to.
    short transpose, rowmap;
    int row0, kvalid;
    int max_steps;                       // > 0: only the first max_steps steps (the ring tail: a copy of a stream's head)
    // kind 1: a head-bias tile instead - dstB[h][r] = (W ++ b)[r] for r < out_dim + in_dim (W: out_dim floats, b: in_dim floats)
    short kind;
    unsigned char ktype[10];
    short kbase[10];
};

// A whole net's segments in ONE launch (round 4).  A repack used to be ~16 tiny launches per stream; a training step repacks the
// forward and the backward stream of both nets after every optimizer step - ~60 launches of a few microseconds each, most of
// the step's "launch gaps" (profiles/r04/train_step_timeline.md).  first_block: prefix sums of the segments' 256-thread blocks.
#define PACK_MAX_SEGS 26
struct PackPlan { int n, Lp, Ld, Lt; int first_block[PACK_MAX_SEGS + 1]; PackSeg seg[PACK_MAX_SEGS]; };
static_assert(sizeof(PackPlan) <= 3800, "a pack plan must fit the kernel-argument segment");

__global__ void __launch_bounds__(256) pack_plan_kernel(PackPlan P) {
    int k = 0;
    while (k + 1 < P.n && (int)blockIdx.x >= P.first_block[k + 1]) ++k;        // block-uniform
    const PackSeg& s = P.seg[k];
    const int e = ((int)blockIdx.x - P.first_block[k]) * 256 + threadIdx.x;
    if (s.kind == 1) {
        if (e < SW_BIAS_TILE_FLOATS) {
            const int r = e & 15;
            s.dstB[e] = r < s.out_dim ? s.W[r] : ((r - s.out_dim) < s.in_dim ? s.b[r - s.out_dim] : 0.f);
        }
        return;
    }
    const int nsteps = s.max_steps > 0 ? s.max_steps : s.NT * s.KT * 4;
    if (e < nsteps * SW_STEP_FLOATS) {
        const int step = e / SW_STEP_FLOATS, rem = e % SW_STEP_FLOATS;
        const int lane = rem >> 2, i4 = rem & 3;
        const int n = step / (s.KT * 4), kt = (step >> 2) % s.KT, q = step & 3;
        const int r = 4 * q + i4, i = lane & 31, h = lane >> 5;
        const int row = 32 * n + i;
        int col = -1;
        switch (s.ktype[kt]) {
            case KT_TRUNK: col = sw_frow(r, h); break;
            case KT_POS0: col = sw_pos_col(r, h, P.Lp); break;
            case KT_POS1: col = sw_pos_col(16 + r, h, P.Lp); break;
            case KT_DIR: col = sw_dir_col(r, h, P.Ld); break;
            case KT_TIME: col = sw_time_col(r, h, P.Lt); break;
        }
        float v = 0.f;
        if (!s.transpose) {
            if (row < s.out_dim && col >= 0) v = s.W[(size_t)row * s.in_dim + s.kbase[kt] + col];
        } else {
            const int kk = s.kbase[kt] + col;
            int wcol = row;
            if (s.rowmap) {
                const int hh = (i >> 2) & 1, rr = (i & 3) + 4 * (i >> 3);      // inverse of sw_frow
                wcol = sw_pos_col(16 * n + rr, hh, P.Lp);
            }
            if (wcol >= 0 && wcol < s.out_dim && kk < s.kvalid) v = s.W[(size_t)kk * s.in_dim + s.row0 + wcol];
        }
        s.dstW[e] = v;
    }
    if (s.b && s.max_steps == 0 && e < s.NT * SW_BIAS_TILE_FLOATS) {
        const int n = e / SW_BIAS_TILE_FLOATS, rem = e % SW_BIAS_TILE_FLOATS;
        const int h = rem >> 4, r = rem & 15;
        const int row = 32 * n + sw_frow(r, h);
        s.dstB[e] = (row < s.out_dim) ? s.b[row] : 0.f;
    }
}

// feature_linear folded into views_linears.0 (swnerf_common.h, SW_CANON_STEPS): fold[u][i] = sum_o Wv[u][o] W_f[o][i] for
// i < 256, fold[u][256 + c] = Wv[u][256 + c] for the Cdir view-direction columns, and behind the matrix
// b_vf[u] = sum_o Wv[u][o] b_f[o] + b_v[u].  Double accumulation, one rounding: the folded row is as close to the exact
// product as a float can be (the two-layer form rounds `feature` AND the second product).  Block u, thread i.
// Reference: model.py:49-53 (feature_linear -> cat -> views_linears[0], no activation in between).
__global__ void __launch_bounds__(1024) fold_views_kernel(const float* Wv, int ldv, const float* bv, const float* Wf, const float* bf,
                                                          int Cdir, float* fold) {
    // one output row u per workgroup; the 256-term sum is dealt over four quarters of the workgroup (a training step folds four
    // times - two nets, forward and backward stream - and a 256-long dependent fp64 chain per thread cost 24 us a time)
    __shared__ float wrow[256];
    __shared__ double part[4][256];
    __shared__ double bprod[256];
    const int u = blockIdx.x, i = threadIdx.x & 255, q = threadIdx.x >> 8;
    if (q == 0) wrow[i] = Wv[(size_t)u * ldv + i];
    __syncthreads();
    double acc = 0.0;
    for (int o = 64 * q; o < 64 * q + 64; ++o) acc += (double)wrow[o] * (double)Wf[(size_t)o * 256 + i];
    part[q][i] = acc;
    if (q == 1) bprod[i] = (double)wrow[i] * (double)bf[i];
    __syncthreads();
    float* row = fold + (size_t)u * SW_FOLD_LD;
    if (q == 0) {
        row[i] = (float)(((part[0][i] + part[1][i]) + part[2][i]) + part[3][i]);
        if (i < SW_FOLD_LD - 256) row[256 + i] = i < Cdir ? Wv[(size_t)u * ldv + 256 + i] : 0.f;
    } else if (q == 1 && i < 64) {                         // b_vf: 64 lanes x 4 terms, then a wave reduction (fp64 throughout, one rounding)
        double b = ((bprod[i] + bprod[64 + i]) + bprod[128 + i]) + bprod[192 + i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) b += __shfl_xor(b, m, 64);
        if (i == 0) fold[(size_t)128 * SW_FOLD_LD + u] = (float)(b + (double)bv[u]);
    }
}

// views_linears.0 . feature_linear of the net `params` (canonical order) -> fold [SW_FOLD_FLOATS]
static int fold_views(const float* const* params, int Cdir, float* fold, hipStream_t st) {
    hipLaunchKernelGGL(fold_views_kernel, dim3(128), dim3(1024), 0, st, params[16], 256 + Cdir, params[17], params[18], params[19], Cdir, fold);
    return sw_check(hipGetLastError(), "pack_net fold launch");
}

// Collects the segments of a stream (weights advance `w`, bias-style tiles advance `b`) and launches them as ONE kernel:
// flush() (also called when the plan is full).  Nothing is launched before that - order writes that other kernels read
// (the fold) IN FRONT of the first flush on the same stream.
struct Packer {
    hipStream_t st; float* w; float* b; int Lp, Ld, Lt; int rc;
    PackPlan plan;
    PackSeg head; bool have_head;                        // the stream's first weight segment (its first SW_TAIL steps are the ring tail)
    Packer(hipStream_t st_, float* w_, float* b_, int Lp_, int Ld_, int Lt_) : st(st_), w(w_), b(b_), Lp(Lp_), Ld(Ld_), Lt(Lt_), rc(0), have_head(false) {
        plan.n = 0; plan.first_block[0] = 0;
    }
    void push(const PackSeg& s, int threads) {
        if (rc) return;
        if (plan.n == PACK_MAX_SEGS) flush();
        plan.seg[plan.n] = s;
        plan.first_block[plan.n + 1] = plan.first_block[plan.n] + (threads + 255) / 256;
        ++plan.n;
    }
    int flush() {
        if (rc || plan.n == 0) return rc;
        plan.Lp = Lp; plan.Ld = Ld; plan.Lt = Lt;
        hipLaunchKernelGGL(pack_plan_kernel, dim3(plan.first_block[plan.n]), dim3(256), 0, st, plan);
        rc = sw_check(hipGetLastError(), "pack_net launch");
        plan.n = 0; plan.first_block[0] = 0;
        return rc;
    }
    static PackSeg blank() {
        PackSeg s;
        s.W = nullptr; s.b = nullptr; s.dstW = nullptr; s.dstB = nullptr; s.out_dim = 0; s.in_dim = 0; s.NT = 0; s.KT = 0;
        s.transpose = 0; s.rowmap = 0; s.row0 = 0; s.kvalid = 0; s.max_steps = 0; s.kind = 0;
        for (int i = 0; i < 10; ++i) { s.ktype[i] = 0; s.kbase[i] = 0; }
        return s;
    }
    void seg(const float* W, const float* bias, int out_dim, int in_dim, int NT, int KT, const int* kt, const int* kb) {
        PackSeg s = blank();
        s.W = W; s.b = bias; s.out_dim = out_dim; s.in_dim = in_dim; s.NT = (short)NT; s.KT = (short)KT;
        for (int i = 0; i < KT; ++i) { s.ktype[i] = (unsigned char)kt[i]; s.kbase[i] = (short)kb[i]; }
        s.dstW = w; s.dstB = b;
        if (!have_head) { head = s; have_head = true; }
        push(s, NT * KT * 4 * SW_STEP_FLOATS);
        w += NT * KT * 4 * SW_STEP_FLOATS;
        if (bias) b += NT * SW_BIAS_TILE_FLOATS;
    }
    // transposed segment of the backward stream: out rows = W columns [row0, row0+n_out), k = W rows [0, kvalid)
    void segT(const float* W, int w_rows, int w_cols, int row0, int n_out, int NT, int KT, int rowmap = 0) {
        PackSeg s = blank();
        s.W = W; s.out_dim = n_out; s.in_dim = w_cols; s.NT = (short)NT; s.KT = (short)KT;
        for (int i = 0; i < KT; ++i) { s.ktype[i] = KT_TRUNK; s.kbase[i] = (short)(32 * i); }
        s.dstW = w; s.dstB = b;
        s.transpose = 1; s.row0 = row0; s.kvalid = w_rows; s.rowmap = (short)rowmap;
        if (!have_head) { head = s; have_head = true; }
        push(s, NT * KT * 4 * SW_STEP_FLOATS);
        w += NT * KT * 4 * SW_STEP_FLOATS;
    }
    // the ring tail: the first SW_TAIL steps of the stream that starts with segment `first` once more, at the current `w`
    // (the head segment must hold at least SW_TAIL steps: every stream here starts with one of >= 16)
    void tail() {
        if (rc) return;
        if (!have_head || head.NT * head.KT * 4 < SW_TAIL) { rc = sw_fail(SWNERF_E_ARG, "pack_net: a stream's head segment is shorter than its ring tail"); return; }
        PackSeg s = head;
        s.b = nullptr; s.dstW = w; s.max_steps = SW_TAIL;
        push(s, SW_TAIL * SW_STEP_FLOATS);
        w += SW_TAIL * SW_STEP_FLOATS;
    }
    // `nout` weight rows of length `in_dim` (a multiple of 32) as bias-style tiles, for head_valu
    void vecs(const float* W, int nout, int in_dim) {
        for (int o = 0; o < nout; ++o) btiles(W + (size_t)o * in_dim, in_dim, in_dim / 32);
    }
    // NT bias tiles of `bias` [out_dim] alone (no weight steps)
    void btiles(const float* bias, int out_dim, int NT) {
        PackSeg s = blank();
        s.W = bias; s.b = bias; s.out_dim = out_dim; s.in_dim = out_dim; s.NT = (short)NT; s.KT = 0;
        s.dstW = w; s.dstB = b;
        push(s, NT * SW_BIAS_TILE_FLOATS);
        b += NT * SW_BIAS_TILE_FLOATS;
    }
    void headbias(const float* b_a, int n_a, const float* b_b, int n_b) {
        PackSeg s = blank();
        s.kind = 1; s.W = b_a; s.b = b_b; s.out_dim = n_a; s.in_dim = n_b; s.dstB = b;
        push(s, SW_BIAS_TILE_FLOATS);
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
        if (Ctime) {
            // layer 0 of the deformation net on cat[gamma(x), gamma(t)] as TWO segments: TIME = bias + the gamma(t) columns (the
            // fused passes evaluate it once per ray, mlp_core.h time_bias_tile; the per-row kernels run it in line), then the
            // gamma(x) columns, accumulating.  A stream that begins here loops back to the gamma(x) segment: TIME is a prefix.
            const int et[1] = {KT_TIME}, ebt[1] = {Cpos};
            const bool first = !have_head;
            seg(P[0], P[1], 256, Cpos + Ctime, 8, 1, et, ebt);
            if (first) have_head = false;
            seg(P[0], nullptr, 256, Cpos + Ctime, 8, 2, e3, eb);
        } else {
            seg(P[0], P[1], 256, Cpos, 8, 2, e3, eb);                           // layer 0
        }
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
    int b8[8];
    for (int i = 0; i < 8; ++i) b8[i] = 32 * i;
    // DIR: the gamma(d) columns of the folded view layer + b_vf, evaluated once per ray by the fused passes (swnerf_common.h)
    auto dir = [&](Packer& pk) {
        pk.seg(fold, fold + 128 * SW_FOLD_LD, 128, SW_FOLD_LD, 4, 1, vt + 8, vb + 8);
        pk.have_head = false;                                                      // the ring tail repeats the head of MAIN, not DIR
    };
    auto canon = [&](Packer& pk) {
        pk.trunk(params, params[20], params[21], 1, Cpos, 0, params[23], 3);       // ... alpha_linear + head biases
        pk.seg(fold, nullptr, 128, SW_FOLD_LD, 4, 8, vt, b8);                      // VIEWSH on h7 (accumulators start from the per-ray tile)
        pk.vecs(params[22], 3, 128);                                               // rgb_linear.weight
    };
    if (kind == SWNERF_NET_DNERF) {
        Packer pk(st, packed, packed + SW_DNERF_W_FLOATS, L_pos, L_dir, L_time);
        dir(pk);
        pk.trunk(params + 24, params[40], params[41], 3, Cpos, Ctime);             // deformation net
        canon(pk);
        if (!pk.rc && pk.w != packed + (size_t)(SW_DEFORM_STEPS + SW_CANON_STEPS) * SW_STEP_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net: internal layout mismatch");
        pk.tail();
        if ((rc = pk.flush())) return rc;
        packed += SW_DNERF_A_FLOATS;                                               // then the canon-only blob
    }
    Packer pk(st, packed, packed + SW_CANON_W_FLOATS, L_pos, L_dir, L_time);
    dir(pk);
    canon(pk);
    if (!pk.rc && (pk.w != packed + (size_t)SW_CANON_STEPS * SW_STEP_FLOATS || pk.b != packed + SW_CANON_VL_OFFSET))
        return sw_fail(SWNERF_E_ARG, "pack_net: internal layout mismatch");
    pk.tail();
    if ((rc = pk.flush())) return rc;
    // the view branch once more, as a stream that wraps onto itself (biases: the tiles packed above)
    Packer vl(st, packed + SW_CANON_VL_OFFSET, nullptr, L_pos, L_dir, L_time);
    int vt2[9], vb2[9];                                                          // k-tile order [gamma(d) | h7]: the summation order of
    vt2[0] = KT_DIR; vb2[0] = 256;                                                // DIR + VIEWSH (mlp_core.h canon_tail_rows)
    for (int i = 0; i < 8; ++i) { vt2[1 + i] = KT_TRUNK; vb2[1 + i] = 32 * i; }
    vl.seg(fold, nullptr, 128, SW_FOLD_LD, 4, 9, vt2, vb2);
    vl.tail();
    return vl.flush();
}

// The canonical net's bias / head tiles in the order the bf16x3 core consumes them (SW_X3_CANON_BIAS_TILES): the fp32 stream's
// tiles with b_vf behind the head-bias tile instead of in front (that core has no per-ray DIR prefix).  `fold`: the folded matrix of
// the fp32 blob made from the same tensors (b_vf sits behind its 128 rows).  Called by swnerf_pack_net_x3_kind (x3_kernels.hip).
int sw_pack_canon_bias_x3(const float* const* params, const float* fold, float* dst, hipStream_t st) {
    Packer pk(st, nullptr, dst, 0, 0, 0);
    for (int l = 0; l < 8; ++l) pk.btiles(params[2 * l + 1], 256, 8);                  // pts_linears.l.bias
    pk.vecs(params[20], 1, 256);                                                       // alpha_linear.weight
    pk.headbias(params[21], 1, params[23], 3);                                         // [b_alpha, b_r, b_g, b_b]
    pk.btiles(fold + 128 * SW_FOLD_LD, 128, 4);                                        // b_vf = Wv[:, :256] . b_f + b_v
    pk.vecs(params[22], 3, 128);                                                       // rgb_linear.weight
    if (!pk.rc && pk.b != dst + SW_X3_CANON_BIAS_TILES * SW_BIAS_TILE_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net_x3: bias layout mismatch");
    return pk.flush();
}

// use_viewdirs=False (model.py:59-60): the trunk alone; output_linear's rows ride as bias-style tiles like the other heads.
extern "C" int swnerf_pack_net_noview(const float* const* params, int L_pos, int out_ch, float* packed, void* stream) {
    if (!params || !packed) return sw_fail(SWNERF_E_ARG, "pack_net_noview: NULL pointer");
    if (L_pos < 0 || L_pos > 10) return sw_fail(SWNERF_E_UNSUPP, "pack_net_noview: %d position bands exceed 10", L_pos);
    if (out_ch < 4 || out_ch > SW_NOVIEW_MAX_OUT)
        return sw_fail(SWNERF_E_UNSUPP, "pack_net_noview: output_ch %d (the reference builds 4 or 5, nerf/run.py:231)", out_ch);
    for (int i = 0; i < 18; ++i) if (!params[i]) return sw_fail(SWNERF_E_ARG, "pack_net_noview: params[%d] is NULL", i);
    hipStream_t st = (hipStream_t)stream;
    Packer pk(st, packed, packed + SW_NOVIEW_W_FLOATS, L_pos, 0, 0);
    pk.trunk(params, params[16], params[17], out_ch, 3 * (1 + 2 * L_pos), 0);
    if (!pk.rc && (pk.w != packed + (size_t)SW_NOVIEW_STEPS * SW_STEP_FLOATS || pk.b != packed + SW_NOVIEW_W_FLOATS + SW_NOVIEW_BIAS_TILES(out_ch) * SW_BIAS_TILE_FLOATS))
        return sw_fail(SWNERF_E_ARG, "pack_net_noview: internal layout mismatch");
    pk.tail();
    return pk.flush();
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
        pk.tail();
        return pk.rc;
    };
    if (bwd_kind == SWNERF_BWD_DEFORM) {
        Packer pk(st, packed_bwd, packed_bwd + SW_DBWD_W_FLOATS, L_pos, L_dir, 0);
        for (int l = 7; l >= 1; --l)                                  // _time.l.weight[:, -256:]^T
            pk.segT(params[2 * l], 256, l == 5 ? Cpos + 256 : 256, l == 5 ? Cpos : 0, 256, 8, 8);
        int rc = tail(pk, SW_DBWD_STEPS);
        if (rc) return rc;
        pk.vecs(params[16], 3, 256);                                  // _time_out.weight rows, tile n: [h][r] = w[o][32n + frow(r,h)]
        return pk.flush();
    }
    const bool fused = bwd_kind == SWNERF_BWD_DNERF_FUSED;
    const bool ig = bwd_kind == SWNERF_BWD_CANON_INPUT_GRAD || fused;
    const size_t wfloats = fused ? SW_BWD_DN_W_FLOATS : (ig ? SW_BWD_IG_W_FLOATS : SW_BWD_W_FLOATS);
    Packer pk(st, packed_bwd, packed_bwd + wfloats, L_pos, L_dir, 0);
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
    return pk.flush();
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
    Packer pk(st, packed_bwd, packed_bwd + SW_DBWD_W_FLOATS, L_pos, 0, 0);
    for (int l = 7; l >= 1; --l)                                  // pts_linears.l.weight[:, -256:]^T
        pk.segT(params[2 * l], 256, l == 5 ? Cpos + 256 : 256, l == 5 ? Cpos : 0, 256, 8, 8);
    if (!pk.rc && pk.w != packed_bwd + (size_t)SW_DBWD_STEPS * SW_STEP_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net_bwd_noview: internal layout mismatch");
    pk.tail();
    pk.vecs(params[16], out_ch, 256);                             // output_linear.weight rows, tile n: [h][r] = w[c][32n + frow(r,h)]
    return pk.flush();
}

extern "C" int swnerf_pack_net_bwd(const float* const* params, int L_pos, int L_dir, float* packed_bwd, void* stream) {
    return swnerf_pack_net_bwd_kind(SWNERF_BWD_CANON, params, L_pos, L_dir, packed_bwd, stream);
}
