// mfma_f32_shapes.hip - does the chip hold a different clock under v_mfma_f32_16x16x4_f32 than under v_mfma_f32_32x32x2_f32?
// (MI355X_MICROARCH.md, DVFS give-back (7): for bf16 the 16x16x32 shape delivers 1.12-1.15 x the FLOP/s of 32x32x16 at equal
// cycles per FLOP, on random data.)  Both loops do the same FLOPs per wave: a 64 x 64 output block per wave, operands re-read
// from LDS every step (ds_read_b32, random data), 16 waves per workgroup, one workgroup per CU - the shape of the weight-
// gradient GEMM's main loop without its memory traffic.  Reports wall TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_shapes mfma_f32_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ROWS 32
#define PITCH 272            // floats per LDS row: 256 + 16 so that rows r and r+1 fall into different bank halves

template <bool S16>
__global__ void __launch_bounds__(1024) loop(const float* src, float* out, unsigned long long* clk, int iters) {
    __shared__ float As[ROWS * PITCH], Bs[ROWS * PITCH];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int e = t; e < ROWS * 256; e += 1024) {
        As[(e >> 8) * PITCH + (e & 255)] = src[e];
        Bs[(e >> 8) * PITCH + (e & 255)] = src[ROWS * 256 + e];
    }
    __syncthreads();
    const int o0 = 64 * (w & 3), i0 = 64 * (w >> 2);
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (S16) {
        f32x4 acc[16];
        for (int k = 0; k < 16; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int io = lane & 15, kk = lane >> 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < ROWS / 4; ++st) {          // 4 rows (k) per MFMA
                float a[4], b[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) { a[x] = As[(4 * st + kk) * PITCH + o0 + 16 * x + io]; b[x] = Bs[(4 * st + kk) * PITCH + i0 + 16 * x + io]; }
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[4 * x + y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[4 * x + y], 0, 0, 0);
            }
        }
        for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    } else {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        const int i = lane & 31, hp = lane >> 5;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < ROWS / 2; ++st) {          // 2 rows (k) per MFMA
                const float a0 = As[(2 * st + hp) * PITCH + o0 + i], a1 = As[(2 * st + hp) * PITCH + o0 + 32 + i];
                const float b0 = Bs[(2 * st + hp) * PITCH + i0 + i], b1 = Bs[(2 * st + hp) * PITCH + i0 + 32 + i];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            }
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 1024 + t] = s;
    if (t == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <bool S16>
static void run(const float* src, float* out, unsigned long long* clk, const char* name) {
    const int iters = 6000;                                  // 32 rows x 64 x 64 x 2 FLOP per wave and iteration
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 40; ++k) loop<S16><<<256, 1024>>>(src, out, clk, iters / 20);   // ~2 s of back-to-back launches first
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop<S16><<<256, 1024>>>(src, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int b = 0; b < 256; ++b) ghz += (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
    const double flop = 256.0 * 16 * iters * ROWS * 64 * 64 * 2;
    printf("| %s | %.3f | %.1f | %.3f |\n", name, ms, flop / (ms * 1e-3) / 1e12, ghz / 256);
}

int main() {
    float *src, *out; unsigned long long* clk;
    hipMalloc(&src, 2 * ROWS * 256 * sizeof(float)); hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&clk, 512 * sizeof(unsigned long long));
    float h[2 * ROWS * 256];
    srand(7);
    for (int k = 0; k < 2 * ROWS * 256; ++k) h[k] = (float)rand() / RAND_MAX - 0.5f;       // random data: zeros would run at full clock
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    printf("| fp32 MFMA shape (64 x 64 block per wave, 16 waves per CU, operands from LDS, random data) | ms | TFLOP/s | in-kernel clock GHz |\n|---|---|---|---|\n");
    for (int rep = 0; rep < 2; ++rep) {
        run<false>(src, out, clk, "v_mfma_f32_32x32x2_f32");
        run<true>(src, out, clk, "v_mfma_f32_16x16x4_f32");
    }
    return 0;
}
