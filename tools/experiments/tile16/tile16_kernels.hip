// tile16_kernels.hip - EXPERIMENT (round 2), not part of libswnerf_hip.so: the static net's MLP on
// 16-row tiles (mlp_core16.h: v_mfma_f32_16x16x4_f32, 182 registers per wave), two wavefronts per SIMD, each with its
// own LDS-DMA weight ring.  Result on MI355X (README.md here): correct (1e-5 of the 32-row kernel) and AS FAST as the
// shipped 32-row / one-wave design (146.7 vs 147.8 TFLOP/s at 786 432 rows), not faster - the shipped kernel already
// sits at the clock-limited matrix rate.  Build: ./build.sh ; run: python probe.py (on the GPU box).
#include <hip/hip_runtime.h>
#include "../../../include/swnerf.h"
#include "swnerf_common.h"
#include "mlp_core16.h"
#include "host_util.h"

static thread_local char g_err16[SW_ERRBUF_LEN];
char* sw_errbuf() { return g_err16; }

// ------------------------------------------------------------------------------------------------------------------
// state_dict -> the 16-row weight stream.  Segment (NT tiles of 16 outputs, KT k-tiles of 16): step = (np, kt, half),
// lane (i = l&15, kq = l>>4), element e: W[16(2np + (e&1)) + i][kbase[kt] + col(kt, q = 2half + (e>>1), kq)].
enum { K16_TRUNK = 0, K16_POS = 1, K16_DIR = 2 };
struct Pack16Seg {
    const float* W; int out_dim, in_dim, NT, KT;
    int ktype[18], kbase[18], klocal[18];
    int Lp, Ld;
    float* dst;
};

__global__ void __launch_bounds__(256) pack16_seg_kernel(Pack16Seg s) {
    const int nsteps = (s.NT / 2) * s.KT * 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nsteps * SW_STEP_FLOATS) return;
    const int step = idx / SW_STEP_FLOATS, rem = idx % SW_STEP_FLOATS;
    const int lane = rem >> 2, e = rem & 3;
    const int np = step / (s.KT * 2), kt = (step >> 1) % s.KT, half = step & 1;
    const int i = lane & 15, kq = lane >> 4;
    const int n = 2 * np + (e & 1), q = 2 * half + (e >> 1);
    const int row = 16 * n + i;
    int col = -1;
    switch (s.ktype[kt]) {
        case K16_TRUNK: col = 4 * kq + q; break;
        case K16_POS: col = sw16_pos_col(4 * s.klocal[kt] + q, kq, s.Lp); break;
        case K16_DIR: col = sw16_dir_col(4 * s.klocal[kt] + q, kq, s.Ld); break;
    }
    s.dst[idx] = (row < s.out_dim && col >= 0) ? s.W[(size_t)row * s.in_dim + s.kbase[kt] + col] : 0.f;
}

__global__ void __launch_bounds__(256) copy_pad_kernel(float* dst, const float* src, int n, int npad) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < npad) dst[e] = e < n ? src[e] : 0.f;
}

extern "C" size_t tile16_packed_floats(void) { return (size_t)SW16_FLOATS; }

extern "C" int tile16_pack_net(const float* const* params, int L_pos, int L_dir, float* packed, void* stream) {
    if (!params || !packed) return sw_fail(SWNERF_E_ARG, "pack_net16: NULL pointer");
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "pack_net16: embedder bands (%d,%d) exceed (10,4)", L_pos, L_dir);
    for (int i = 0; i < 24; ++i) if (!params[i]) return sw_fail(SWNERF_E_ARG, "pack_net16: params[%d] is NULL", i);
    const int Cpos = 3 * (1 + 2 * L_pos), Cdir = 3 * (1 + 2 * L_dir);
    hipStream_t st = (hipStream_t)stream;
    float* w = packed;
    int rc = 0;
    auto seg = [&](const float* W, int out_dim, int in_dim, int NT, int KT, int ntrunk, int trunk_base, int nemb, int etype, int ebase) {
        if (rc) return;
        Pack16Seg s;
        s.W = W; s.out_dim = out_dim; s.in_dim = in_dim; s.NT = NT; s.KT = KT; s.Lp = L_pos; s.Ld = L_dir; s.dst = w;
        for (int k = 0; k < 18; ++k) { s.ktype[k] = 0; s.kbase[k] = 0; s.klocal[k] = 0; }
        for (int k = 0; k < ntrunk; ++k) { s.ktype[k] = K16_TRUNK; s.kbase[k] = trunk_base + 16 * k; }
        for (int k = 0; k < nemb; ++k) { s.ktype[ntrunk + k] = etype; s.kbase[ntrunk + k] = ebase; s.klocal[ntrunk + k] = k; }
        const int total = (NT / 2) * KT * 2 * SW_STEP_FLOATS;
        hipLaunchKernelGGL(pack16_seg_kernel, dim3((total + 255) / 256), dim3(256), 0, st, s);
        rc = sw_check(hipGetLastError(), "pack_net16 launch");
        w += total;
    };
    seg(params[0], 256, Cpos, 16, 4, 0, 0, 4, K16_POS, 0);                              // L0
    for (int l = 1; l < 8; ++l) {
        if (l == 5) {
            seg(params[10], 256, Cpos + 256, 16, 16, 16, Cpos, 0, 0, 0);                // L5, h part
            seg(params[10], 256, Cpos + 256, 16, 4, 0, 0, 4, K16_POS, 0);               // L5, gamma(x) part
        } else {
            seg(params[2 * l], 256, 256, 16, 16, 16, 0, 0, 0, 0);
        }
    }
    seg(params[18], 256, 256, 16, 16, 16, 0, 0, 0, 0);                                  // FEAT
    seg(params[16], 128, 256 + Cdir, 8, 18, 16, 0, 2, K16_DIR, 256);                    // VIEWS
    if (rc) return rc;
    if (w != packed + (size_t)SW_CANON_STEPS * SW_STEP_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net16: internal layout mismatch");
    rc = sw_check(hipMemcpyAsync(w, packed, (size_t)SW_TAIL * SW_STEP_FLOATS * sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net16 tail copy");
    if (rc) return rc;
    float* b = packed + SW16_W_FLOATS;
    auto vec = [&](const float* src, int n, int npad) {
        if (rc) return;
        hipLaunchKernelGGL(copy_pad_kernel, dim3((npad + 255) / 256), dim3(256), 0, st, b, src, n, npad);
        rc = sw_check(hipGetLastError(), "pack_net16 launch");
        b += npad;
    };
    for (int l = 0; l < 8; ++l) vec(params[2 * l + 1], 256, 256);                       // pts_linears biases
    vec(params[20], 256, 256);                                                          // alpha_linear.weight
    if (!rc) {                                                                          // head biases [b_alpha, b_r, b_g, b_b, 0...]
        rc = sw_check(hipMemsetAsync(b, 0, 16 * sizeof(float), st), "pack_net16 memset");
        if (!rc) rc = sw_check(hipMemcpyAsync(b, params[21], sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net16 copy");
        if (!rc) rc = sw_check(hipMemcpyAsync(b + 1, params[23], 3 * sizeof(float), hipMemcpyDeviceToDevice, st), "pack_net16 copy");
        b += 16;
    }
    vec(params[19], 256, 256);                                                          // feature_linear.bias
    vec(params[17], 128, 128);                                                          // views_linears.0.bias
    vec(params[22], 384, 384);                                                          // rgb_linear.weight [3,128]
    if (!rc && b != packed + SW16_FLOATS) return sw_fail(SWNERF_E_ARG, "pack_net16: internal bias layout mismatch");
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// model.forward(x) on already-embedded rows, 16 rows per wave; 4 waves per workgroup, TWO workgroups per CU (2 waves
// per SIMD from different workgroups: one's prologue / epilogue runs under the other's MFMAs).
#define R16_LDS_WAVE_FLOATS (SW_RING * SW_STEP_FLOATS)
struct Mlp16Dev { const float* x; int64_t M; int C, Cpos, Lp, Ld; const float* w0; const float* b0; float* out; };

__global__ void __launch_bounds__(256, 2) mlp_forward16_kernel(Mlp16Dev P) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x * 4; i < SW16_BIAS_FLOATS; i += 256 * 4)
        *reinterpret_cast<f32x4*>(lds_all + i) = *reinterpret_cast<const f32x4*>(P.b0 + i);
    __syncthreads();
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
    if (tile * 16 >= P.M) return;
    const int64_t row = tile * 16 + j;
    const bool live = row < P.M;
    const float* xr = P.x + (live ? row : P.M - 1) * P.C;
    float* lds_ring = lds_all + SW16_BIAS_FLOATS + wv * R16_LDS_WAVE_FLOATS;
    f32x4 emb[4], demb[2], in[16], out[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) { const int col = sw16_pos_col(s, g, P.Lp); emb[s >> 2][s & 3] = col >= 0 ? xr[col] : 0.f; }
#pragma unroll
    for (int s = 0; s < 8; ++s) { const int col = sw16_dir_col(s, g, P.Ld); demb[s >> 2][s & 3] = col >= 0 ? xr[P.Cpos + col] : 0.f; }
    WStream ws;
    ws_start(ws, P.w0, lds_all, lds_ring, lane);
    ws.bias = lds_all + 4 * g;
    float sigma, rgb[3];
    trunk16(emb, in, out, sigma, ws, 4 * g);
    tail16(in, out, demb, rgb, lds_all + 8 * 256 + 256, ws);
    if (live && g == 0) {
        f32x4 r4 = {rgb[0], rgb[1], rgb[2], sigma};
        *reinterpret_cast<f32x4*>(P.out + row * 4) = r4;
    }
}

extern "C" int tile16_mlp_forward(const float* packed16, const float* x, int64_t M, int L_pos, int L_dir, float* out, void* stream) {
    if (M == 0 && packed16) return 0;
    if (!packed16 || !x || !out || M < 0) return sw_fail(SWNERF_E_ARG, "mlp_forward16: NULL pointer or negative M");
    if (L_pos < 0 || L_pos > 10 || L_dir < 0 || L_dir > 4) return sw_fail(SWNERF_E_UNSUPP, "mlp_forward16: embedder bands (%d,%d) exceed (10,4)", L_pos, L_dir);
    Mlp16Dev P;
    P.x = x; P.M = M; P.Lp = L_pos; P.Ld = L_dir; P.Cpos = 3 * (1 + 2 * L_pos); P.C = P.Cpos + 3 * (1 + 2 * L_dir);
    P.w0 = packed16; P.b0 = packed16 + SW16_W_FLOATS; P.out = out;
    const size_t lds = (SW16_BIAS_FLOATS + 4 * R16_LDS_WAVE_FLOATS) * sizeof(float);
    hipLaunchKernelGGL(mlp_forward16_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), lds, (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), "mlp_forward16 launch");
}
