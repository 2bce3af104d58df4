// two_waves_ring.hip - the structural experiment of VERDICT r01 #7 as a microbenchmark: can TWO waves per SIMD, each with
// 16-row tiles (v_mfma_f32_16x16x4_f32, half the registers of the 32-row design) and its OWN LDS-DMA weight ring, keep the
// matrix pipe busier than ONE wave per SIMD with v_mfma_f32_32x32x2_f32 (the shipped structure, where VALU work of a wave
// is not hidden behind its own MFMAs: mfma_shadow.hip)?  The weight stream per FLOP doubles with 16-row tiles (the same
// 1 KiB step feeds 4 MFMAs of half the size), so the question is whether L2 -> LDS delivery keeps up.
// Each wave: ring of 8 x 1 KiB slots filled by global_load_lds_dwordx4 from a 2.4 MB buffer (L2 resident, wraps), per step
// wait -> ds_read_b128 (one step ahead) -> 4 dependent MFMAs -> refill -> V unrelated VALU ops (the non-MFMA work).
// Build: hipcc --offload-arch=gfx950 -O3 -I ../../sw-nerf_amd/csrc -o two_waves_ring two_waves_ring.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lds_dma.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RING 8
#define STREAM_STEPS 2320                      // one net: 2.38 MB

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <bool M16, int V, int WPB>
__global__ void __launch_bounds__(WPB * 64) probe(const float* w, float* out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* ring = lds + wv * RING * 256;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);
    const char* base = reinterpret_cast<const char*>(w);
    const unsigned voff = lane * 16u;
    f32x16 acc32[2] = {};
    f32x4 acc16[8] = {};
    float b = 1.0001f, v0 = lane, v1 = lane + 1.f, v2 = lane + 2.f, v3 = lane + 3.f;
#pragma unroll
    for (int s = 0; s < RING; ++s) ws_dma(base + s * 1024, voff, lds0 + s * 1024);
    wait_vm<RING - 1>();
    f32x4 a = *reinterpret_cast<const f32x4*>(ring + lane * 4);
    for (int t = 0; t < tiles; ++t) {
#pragma unroll 1
        for (int s0 = 0; s0 < STREAM_STEPS; s0 += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                wait_vm<RING - 2>();
                const f32x4 an = *reinterpret_cast<const f32x4*>(ring + ((u + 1) % RING) * 256 + lane * 4);
                __builtin_amdgcn_sched_barrier(0);
                if (M16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc16[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b, acc16[u], 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc32[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b, acc32[u & 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                int nxt = s0 + u + RING;                                       // wraps: the stream is re-read tile after tile
                nxt = __builtin_amdgcn_readfirstlane(nxt >= STREAM_STEPS ? nxt - STREAM_STEPS : nxt);
                ws_dma(base + nxt * 1024, voff, lds0 + u * 1024);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    if ((k & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(b));
                    if ((k & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v1) : "v"(b));
                    if ((k & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v2) : "v"(b));
                    if ((k & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v3) : "v"(b));
                }
                __builtin_amdgcn_sched_barrier(0);
                a = an;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += acc32[0][r] + acc32[1][r];
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 4; ++r) s += acc16[n][r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool M16, int V, int WPB>
static void run(const float* w, float* out, const char* what) {
    // WPB = 4: one wave per SIMD (256-thread workgroups, LDS sized so that only one fits a CU); WPB = 8: two per SIMD
    const size_t ldsb = WPB == 4 ? 100 * 1024 : (size_t)WPB * RING * 1024;
    const int tiles = 12, grid = 1024;          // 1024 workgroups on 256 CUs: 4 rounds
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<M16, V, WPB>), dim3(grid), dim3(WPB * 64), ldsb, 0, w, out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<M16, V, WPB>), dim3(grid), dim3(WPB * 64), ldsb, 0, w, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // matrix-pipe cycles per SIMD: rounds x waves per SIMD x tiles x steps x 4 MFMAs x (32 | 64) cycles
    const double cyc = 4.0 * (WPB / 4) * tiles * STREAM_STEPS * 4 * (M16 ? 32 : 64);
    const double ideal_ms = cyc / 2.4e6;
    const double bytes_per_clk_cu = (double)WPB * 1024 / (4.0 * (M16 ? 32 : 64) * (WPB / 4));
    printf("| %s | %d | %d | %8.3f | %8.3f | %5.1f %% | %4.0f |\n", what, WPB / 4, V, ms, ideal_ms, 100.0 * ideal_ms / ms, bytes_per_clk_cu);
}

int main() {
    float *w, *out;
    hipMalloc(&w, (STREAM_STEPS + 16) * 1024);
    hipMemset(w, 0, (STREAM_STEPS + 16) * 1024);
    hipMalloc(&out, 1024 * 512 * sizeof(float));
    printf("| MFMA / tile rows | waves per SIMD | VALU ops per step | ms | ms at 2.4 GHz, pipe always busy | matrix pipe busy (of nominal clock) | weight stream B/clk/CU |\n|---|---|---|---|---|---|---|\n");
    run<false, 0, 4>(w, out, "32x32x2 (32 rows)");
    run<false, 2, 4>(w, out, "32x32x2 (32 rows)");
    run<false, 4, 4>(w, out, "32x32x2 (32 rows)");
    run<false, 2, 8>(w, out, "32x32x2 (hypothetical: 2 waves would need 2 x 444 registers)");
    run<true, 0, 8>(w, out, "16x16x4 (16 rows)");
    run<true, 2, 8>(w, out, "16x16x4 (16 rows)");
    run<true, 4, 8>(w, out, "16x16x4 (16 rows)");
    run<true, 2, 4>(w, out, "16x16x4 (16 rows)");
    return 0;
}
