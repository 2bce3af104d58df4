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

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

static int generic_launch(const GenericGemm& P, bool bt, void* stream, const char* what) {
    if (P.M == 0) return 0;
    if (!P.A || !P.B || !P.C || P.M < 0 || P.N < 1 || P.K < 1 || P.lda < P.K || P.ldc < P.N || P.ldb < (bt ? P.K : P.N))
        return sw_fail(SWNERF_E_ARG, "%s: bad arguments (M=%lld N=%d K=%d lda=%d ldb=%d ldc=%d)", what, (long long)P.M, P.N, P.K, P.lda, P.ldb, P.ldc);
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
