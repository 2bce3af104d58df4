// wave_dpp.h - wavefront scans and reductions on the DPP data path (gfx9 / CDNA row shifts and row broadcasts).
//
// The compositing kernels (ray.py:155-198: exclusive cumprod, sums over the samples of a ray) are VALU-issue bound once
// their loads are coalesced: a wave64 VALU instruction occupies its SIMD for 4 cycles, and a Hillis-Steele scan through
// __shfl_up costs per step two ds_bpermute (a double is two dwords), their address arithmetic, two selects and the
// multiply.  On the DPP path a step is two v_mov_b32_dpp and the multiply - lanes without a source keep the identity that
// `old` carries - and it never touches the LDS pipe.
//   steps: row_shr:1,2,4,8 (inclusive scan inside each row of 16 lanes), row_bcast:15 into rows 1 and 3, row_bcast:31 into
//   rows 2 and 3.  The association differs from a shfl_up scan (rows first, then row totals); in double that is a 1e-16
//   matter, invisible after the result is rounded to float.
// CALL THESE WITH ALL 64 LANES ACTIVE: a lane switched off by EXEC is an invalid DPP source (its readers keep `old`), so a
// call inside a lane-dependent branch or select arm silently drops neighbours - hoist the call, select afterwards.
#pragma once
#include <hip/hip_runtime.h>

#define SW_DPP_ROW_SHR(n) (0x110 + (n))
#define SW_DPP_WAVE_SHL1 0x130          // lane i <- lane i+1
#define SW_DPP_WAVE_SHR1 0x138          // lane i <- lane i-1
#define SW_DPP_ROW_BCAST15 0x142
#define SW_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, BANK_MASK, false));
}

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

// inclusive product over lanes 0..lane of the wave
__device__ __forceinline__ double wave_incl_prod_f64(double v) {
    v *= dpp_f64<SW_DPP_ROW_SHR(1)>(1.0, v);
    v *= dpp_f64<SW_DPP_ROW_SHR(2)>(1.0, v);
    v *= dpp_f64<SW_DPP_ROW_SHR(4)>(1.0, v);
    v *= dpp_f64<SW_DPP_ROW_SHR(8)>(1.0, v);
    v *= dpp_f64<SW_DPP_ROW_BCAST15, 0xa>(1.0, v);
    v *= dpp_f64<SW_DPP_ROW_BCAST31, 0xc>(1.0, v);
    return v;
}

// inclusive sum over lanes 0..lane of the wave
__device__ __forceinline__ double wave_incl_sum_f64(double v) {
    v += dpp_f64<SW_DPP_ROW_SHR(1)>(0.0, v);
    v += dpp_f64<SW_DPP_ROW_SHR(2)>(0.0, v);
    v += dpp_f64<SW_DPP_ROW_SHR(4)>(0.0, v);
    v += dpp_f64<SW_DPP_ROW_SHR(8)>(0.0, v);
    v += dpp_f64<SW_DPP_ROW_BCAST15, 0xa>(0.0, v);
    v += dpp_f64<SW_DPP_ROW_BCAST31, 0xc>(0.0, v);
    return v;
}

// the value of the lane below (lane 0: `first`), the lane above (lane 63: `last`)
__device__ __forceinline__ double wave_from_below_f64(double v, double first) { return dpp_f64<SW_DPP_WAVE_SHR1>(first, v); }
__device__ __forceinline__ float wave_from_above_f32(float v, float last) { return dpp_f32<SW_DPP_WAVE_SHL1>(last, v); }

// lane 63's value as a wave-uniform
__device__ __forceinline__ double wave_last_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// sum over the wave, valid in lane 63 (the other lanes hold partial sums)
__device__ __forceinline__ float wave_sum_to_last_f32(float v) {
    v += dpp_f32<SW_DPP_ROW_SHR(1)>(0.f, v);
    v += dpp_f32<SW_DPP_ROW_SHR(2)>(0.f, v);
    v += dpp_f32<SW_DPP_ROW_SHR(4)>(0.f, v);
    v += dpp_f32<SW_DPP_ROW_SHR(8)>(0.f, v);
    v += dpp_f32<SW_DPP_ROW_BCAST15, 0xa>(0.f, v);
    v += dpp_f32<SW_DPP_ROW_BCAST31, 0xc>(0.f, v);
    return v;
}
__device__ __forceinline__ double wave_sum_to_last_f64(double v) { return wave_incl_sum_f64(v); }
