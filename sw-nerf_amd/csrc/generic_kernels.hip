// generic_kernels.hip - the NeRF MLPs at ANY shape (model.py:10-62, 93-151, 227-296 with D, W, skips, input sizes and
// use_viewdirs as given): one nn.Linear at a time.  The shipped configs (D=8, W=256, skips=[4], use_viewdirs=True) run
// on the register-resident fused kernels (mlp_core.h); everything else - `use_viewdirs=False` (the reference's argparse
// default, utils.py:26-29 / model.py:59-60), other depths / widths / skip sets - runs layer by layer on this tiled
// fp32-MFMA GEMM, which is slower but exact in the same sense (v_mfma_f32_32x32x2_f32, fp32 accumulate).
//
//   linear   : Y[M,N] = act(X[M,K] . W[N,K]^T + b)         (torch.nn.functional.linear [+ relu])
//   gemm_nn  : dX[M,K] = dY[M,N] . W[N,K]                   (its input gradient)
//   relu_mask: dY *= (Y > 0)                                (relu backward, in place)
// The weight gradient dW = dY^T . X and db = column sums are swnerf_gemm_tn (backward_kernels.hip).
//
// One workgroup (4 waves) owns a 64 x 64 block of the output, wave w the 32 x 32 tile (w&1, w>>1); the K dimension
// is walked in chunks of 32 staged through LDS with bounds-checked 4-byte loads (any leading dimension, any K: 63, 90,
// 319 ...; consecutive lanes read consecutive k of one row, so the loads coalesce).  In the MFMA both operands use
// the same (lane half, step) -> k map, so any k order is a valid contraction order.
#include <hip/hip_runtime.h>
#include "../../include/swnerf.h"
#include "host_util.h"
#include <cstdlib>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GK_CH 32                 // k per chunk
#define GK_LD (GK_CH + 1)        // LDS row pitch (floats): odd, so the 32 rows of a tile hit 32 different banks

struct GenericGemm {
    const float* A; int lda;     // [M, K] row-major
    const float* B; int ldb;     // BT: [N, K] row-major (C = A.B^T);  else [K, N] row-major (C = A.B)
    const float* bias;           // [N] or NULL
    float* C; int ldc;           // [M, N]
    int64_t M; int N, K; int relu;
};

template <bool BT>
__global__ void __launch_bounds__(256) generic_gemm_kernel(GenericGemm P) {
    __shared__ float As[64][GK_LD];
    __shared__ float Bs[64][GK_LD];
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wv = t >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * 64;     // rows on grid.x (2^31-1 blocks); grid.y is capped at 65535
    const int n0 = blockIdx.y * 64;
    const int wm = 32 * (wv & 1), wn = 32 * (wv >> 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < P.K; k0 += GK_CH) {
        // stage A[m0..+64][k0..+32] and op(B)[n0..+64][k0..+32]: 2048 floats each, 8 per thread
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = t + 256 * q, row = e >> 5, kk = e & 31;
            const int64_t m = m0 + row;
            const int k = k0 + kk;
            As[row][kk] = (m < P.M && k < P.K) ? P.A[m * P.lda + k] : 0.f;
        }
        if (BT) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = t + 256 * q, row = e >> 5, kk = e & 31;
                const int n = n0 + row, k = k0 + kk;
                Bs[row][kk] = (n < P.N && k < P.K) ? P.B[(int64_t)n * P.ldb + k] : 0.f;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = t + 256 * q, kk = e >> 6, col = e & 63;       // consecutive lanes: consecutive n of one k row
                const int n = n0 + col, k = k0 + kk;
                Bs[col][kk] = (n < P.N && k < P.K) ? P.B[(int64_t)k * P.ldb + n] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GK_CH / 2; ++s) {
            const float a = As[wm + i][2 * s + h];
            const float b = Bs[wn + i][2 * s + h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map: register r of lane (j = i, h) is row (r&3) + 8(r>>2) + 4h of the tile, column j
    const int n = n0 + wn + i;
    if (n >= P.N) return;
    const float bv = P.bias ? P.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < P.M) {
            float v = acc[r] + bv;
            if (P.relu) v = fmaxf(v, 0.f);
            P.C[m * P.ldc + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same GEMM for outputs at least 64 wide (round 3): a 128 x 128 block of C per workgroup, wave w the 64 x 64 quadrant
// (w&1, w>>1) as 2 x 2 accumulator tiles - four times the MFMA work per staged float of the 64 x 64 kernel above, which ran
// the 8x256 nets at 29 % of the fp32-MFMA roofline.  K is walked in chunks of 32 through two LDS buffers: the next chunk's
// global loads are in flight (registers) while this chunk feeds the matrix pipe, one barrier per chunk.  Operand tiles sit
// K-MAJOR in LDS ([k][row]), so an MFMA operand read is 32 consecutive floats (conflict free) and the pitch is chosen per
// operand so that the staging writes are conflict free too: 129 floats where a thread writes 4 consecutive k of one row
// (A, and B as [N,K]), 132 where it writes 4 consecutive columns of one k (B as [K,N]: one ds_write_b128).
// VEC: both operands allow 16-byte loads (base, leading dimension); otherwise 4-byte loads with the same coalescing.
#define G2_KC 32
#define G2_PA 129
template <bool BT, bool VEC>
__global__ void __launch_bounds__(256, 2) generic_gemm128_kernel(GenericGemm P) {
    constexpr int PB = BT ? 129 : 132;
    extern __shared__ __attribute__((aligned(16))) float g2_lds[];          // [2][ A: KC x PA | B: KC x PB ]
    constexpr int BUF = G2_KC * (G2_PA + PB);
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, h = lane >> 5;
    const int wv = t >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;
    const int wm = 64 * (wv & 1), wn = 64 * (wv >> 1);
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float ra[16], rb[16];                                                     // the next chunk, on its way from global memory
    auto load = [&](int k0) {
        if (VEC) {
            // A (and B^T): thread -> (row = t/8 + 32 q, k = 4 (t%8) ..+3): 8 threads read one 128-byte line of a row
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (t >> 3) + 32 * q, kk = 4 * (t & 7);
                const int64_t m = m0 + row;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < P.M && k0 + kk < P.K) v = *reinterpret_cast<const f32x4*>(P.A + m * P.lda + k0 + kk);   // K % 4 == 0 on this path
                ra[4 * q] = v[0]; ra[4 * q + 1] = v[1]; ra[4 * q + 2] = v[2]; ra[4 * q + 3] = v[3];
            }
            if (BT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = (t >> 3) + 32 * q, kk = 4 * (t & 7);
                    const int n = n0 + row;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (n < P.N && k0 + kk < P.K) v = *reinterpret_cast<const f32x4*>(P.B + (int64_t)n * P.ldb + k0 + kk);
                    rb[4 * q] = v[0]; rb[4 * q + 1] = v[1]; rb[4 * q + 2] = v[2]; rb[4 * q + 3] = v[3];
                }
            } else {
                // B [K,N]: thread -> (k = t/32 + 8 q, n = 4 (t%32) ..+3): 32 threads read 512 consecutive bytes of a k row
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = (t >> 5) + 8 * q, col = 4 * (t & 31);
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (k0 + kk < P.K && n0 + col < P.N) v = *reinterpret_cast<const f32x4*>(P.B + (int64_t)(k0 + kk) * P.ldb + n0 + col);   // N % 4 == 0 on this path
                    rb[4 * q] = v[0]; rb[4 * q + 1] = v[1]; rb[4 * q + 2] = v[2]; rb[4 * q + 3] = v[3];
                }
            }
        } else {
            // 4-byte loads: element e = t + 256 q -> (row = e/32, k = e%32) for A and B^T, (k = e/128, n = e%128) for B [K,N]
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = t + 256 * q, row = e >> 5, kk = e & 31;
                const int64_t m = m0 + row;
                ra[q] = (m < P.M && k0 + kk < P.K) ? P.A[m * P.lda + k0 + kk] : 0.f;
                if (BT) {
                    const int n = n0 + row;
                    rb[q] = (n < P.N && k0 + kk < P.K) ? P.B[(int64_t)n * P.ldb + k0 + kk] : 0.f;
                } else {
                    const int k2 = e >> 7, col = e & 127;
                    rb[q] = (k0 + k2 < P.K && n0 + col < P.N) ? P.B[(int64_t)(k0 + k2) * P.ldb + n0 + col] : 0.f;
                }
            }
        }
    };
    auto stage = [&](int buf) {
        float* As = g2_lds + buf * BUF;
        float* Bs = As + G2_KC * G2_PA;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (t >> 3) + 32 * q, kk = 4 * (t & 7);
#pragma unroll
                for (int e = 0; e < 4; ++e) As[(kk + e) * G2_PA + row] = ra[4 * q + e];
                if (BT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Bs[(kk + e) * PB + row] = rb[4 * q + e];
                }
            }
            if (!BT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = (t >> 5) + 8 * q, col = 4 * (t & 31);
                    const f32x4 v = {rb[4 * q], rb[4 * q + 1], rb[4 * q + 2], rb[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(Bs + kk * PB + col) = v;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = t + 256 * q, row = e >> 5, kk = e & 31;
                As[kk * G2_PA + row] = ra[q];
                if (BT) Bs[kk * PB + row] = rb[q];
                else Bs[(e >> 7) * PB + (e & 127)] = rb[q];
            }
        }
    };
    const int nch = (P.K + G2_KC - 1) / G2_KC;
    load(0);
    stage(0);
    __syncthreads();
#pragma nounroll
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) load((c + 1) * G2_KC);                               // in flight while this chunk computes
        const float* As = g2_lds + (c & 1) * BUF + wm + i;
        const float* Bs = g2_lds + (c & 1) * BUF + G2_KC * G2_PA + wn + i;
#pragma unroll
        for (int s = 0; s < G2_KC / 2; ++s) {
            const float a0 = As[(2 * s + h) * G2_PA], a1 = As[(2 * s + h) * G2_PA + 32];
            const float b0 = Bs[(2 * s + h) * PB], b1 = Bs[(2 * s + h) * PB + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
        if (c + 1 < nch) stage((c + 1) & 1);                                   // the buffer nobody reads in this iteration
        __syncthreads();
    }
    // C/D map: register r of lane (j = i, h) of tile (x, y) is row wm + 32 x + frow(r,h), column wn + 32 y + j
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int n = n0 + wn + 32 * y + i;
            if (n >= P.N) continue;
            const float bv = P.bias ? P.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < P.M) {
                    float v = acc[2 * x + y][r] + bv;
                    if (P.relu) v = fmaxf(v, 0.f);
                    P.C[m * P.ldc + n] = v;
                }
            }
        }
}

static int generic_launch(const GenericGemm& P, bool bt, void* stream, const char* what) {
    if (P.M == 0) return 0;
    if (!P.A || !P.B || !P.C || P.M < 0 || P.N < 1 || P.K < 1 || P.lda < P.K || P.ldc < P.N || P.ldb < (bt ? P.K : P.N))
        return sw_fail(SWNERF_E_ARG, "%s: bad arguments (M=%lld N=%d K=%d lda=%d ldb=%d ldc=%d)", what, (long long)P.M, P.N, P.K, P.lda, P.ldb, P.ldc);
    if (P.N >= 64 && P.M >= 128 && getenv("SWNERF_GENERIC_GEMM_OLD") == nullptr) {
        const int64_t gx2 = (P.M + 127) / 128, gy2 = (P.N + 127) / 128;
        if (gx2 > 0x7fffffffLL || gy2 > 65535) return sw_fail(SWNERF_E_UNSUPP, "%s: M %lld or N %d too large for one launch", what, (long long)P.M, P.N);
        const dim3 grid2((unsigned)gx2, (unsigned)gy2), block2(256);
        const bool vec = (P.lda % 4 == 0) && (P.ldb % 4 == 0) && (((uintptr_t)P.A | (uintptr_t)P.B) % 16 == 0) && (P.K % 4 == 0) && (bt || P.N % 4 == 0);
        const size_t lds = 2 * G2_KC * (G2_PA + (bt ? 129 : 132)) * sizeof(float);
        hipStream_t st = (hipStream_t)stream;
        if (bt && vec) hipLaunchKernelGGL((generic_gemm128_kernel<true, true>), grid2, block2, lds, st, P);
        else if (bt) hipLaunchKernelGGL((generic_gemm128_kernel<true, false>), grid2, block2, lds, st, P);
        else if (vec) hipLaunchKernelGGL((generic_gemm128_kernel<false, true>), grid2, block2, lds, st, P);
        else hipLaunchKernelGGL((generic_gemm128_kernel<false, false>), grid2, block2, lds, st, P);
        return sw_check(hipGetLastError(), what);
    }
    const int64_t gx = (P.M + 63) / 64, gy = (P.N + 63) / 64;
    if (gx > 0x7fffffffLL || gy > 65535) return sw_fail(SWNERF_E_UNSUPP, "%s: M %lld or N %d too large for one launch", what, (long long)P.M, P.N);
    const dim3 grid((unsigned)gx, (unsigned)gy), block(256);
    if (bt) hipLaunchKernelGGL(generic_gemm_kernel<true>, grid, block, 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(generic_gemm_kernel<false>, grid, block, 0, (hipStream_t)stream, P);
    return sw_check(hipGetLastError(), what);
}

extern "C" int swnerf_linear(const float* x, int ldx, int64_t M, int K, const float* weight, const float* bias, int N,
                             int relu, float* y, int ldy, void* stream) {
    GenericGemm P;
    P.A = x; P.lda = ldx; P.B = weight; P.ldb = K; P.bias = bias; P.C = y; P.ldc = ldy; P.M = M; P.N = N; P.K = K; P.relu = relu;
    return generic_launch(P, true, stream, "linear");
}

extern "C" int swnerf_gemm_nn(const float* a, int lda, int64_t M, int K, const float* b, int ldb, int N, float* c, int ldc,
                              void* stream) {
    GenericGemm P;
    P.A = a; P.lda = lda; P.B = b; P.ldb = ldb; P.bias = nullptr; P.C = c; P.ldc = ldc; P.M = M; P.N = N; P.K = K; P.relu = 0;
    return generic_launch(P, false, stream, "gemm_nn");
}

__global__ void __launch_bounds__(256) relu_mask_kernel(float* dy, const float* y, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) dy[e] = (y[e] > 0.f) ? dy[e] : 0.f;
}

extern "C" int swnerf_relu_mask(float* dy, const float* y, int64_t n, void* stream) {
    if (n == 0) return 0;
    if (!dy || !y || n < 0) return sw_fail(SWNERF_E_ARG, "relu_mask: NULL pointer or negative count");
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, n);
    return sw_check(hipGetLastError(), "relu_mask launch");
}
