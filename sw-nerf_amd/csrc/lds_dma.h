// lds_dma.h - global -> LDS copies with no register destination (`global_load_lds_dwordx4`, gfx950).
#pragma once
#include <hip/hip_runtime.h>

// one 1-KiB step: global [gbase + lane*16] -> LDS [lds_addr + lane*16].  M0 carries the LDS
// base; it is compiler-reserved, so it is saved and restored inside the statement.
// The caller orders it after the last LDS read of the destination (write-after-read).
__device__ __forceinline__ void ws_dma(const char* gbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}


// 4 bytes per lane: global [gbase + voff(lane)] -> LDS [lds_addr + lane*4]   (256 B per wave instruction)
__device__ __forceinline__ void lds_dma_dword(const char* gbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}
