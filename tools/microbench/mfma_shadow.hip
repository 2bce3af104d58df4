// mfma_shadow.hip - how many independent VALU instructions fit "for free" between two v_mfma_f32_32x32x2_f32 of ONE wave
// per SIMD (the occupancy of the render kernels)?  Times a loop of dependent MFMAs with K v_fma_f32 on unrelated
// registers after each.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, bool DEP>
__global__ void __launch_bounds__(256, 1) probe(float* out, int iters) {
    f32x16 acc0 = {}, acc1 = {};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (DEP || (u & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {                      // a dependent chain per register, 4 chains interleaved
                if ((k & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(b));
                if ((k & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(b));
                if ((k & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(b));
                if ((k & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(b));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int K, bool DEP>
static void run(float* out) {
    const int iters = 20000;                                  // 160 000 MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<K, DEP><<<1024, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<K, DEP><<<1024, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // 1024 WGs x 4 waves on 1024 SIMDs: 4 rounds of one wave per SIMD
    const double per_mfma_ns = ms * 1e6 / (4.0 * iters * 8);
    printf("| %s | %2d | %8.3f | %6.1f |\n", DEP ? "same accumulator" : "alternating accumulators", K, ms, per_mfma_ns);
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    printf("| MFMA chain | VALU ops after each MFMA | ms | ns per MFMA (64 cycles = 26.7 ns at 2.4 GHz) |\n|---|---|---|---|\n");
    run<0, true>(out); run<2, true>(out); run<4, true>(out); run<8, true>(out); run<12, true>(out); run<16, true>(out); run<24, true>(out);
    run<0, false>(out); run<4, false>(out); run<8, false>(out); run<16, false>(out);
    return 0;
}
