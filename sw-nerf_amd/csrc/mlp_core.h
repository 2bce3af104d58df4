// mlp_core.h - the 8x256 NeRF MLP evaluated by ONE wavefront on a tile of 32 (ray,sample)
// rows with v_mfma_f32_32x32x2_f32, activations resident in registers.
//
// Orientation: the transposed problem  H_out^T[256 x 32] = W[256 x K] . H_in^T[K x 32].
//   A operand = weights (lane (i,h): W[32n+i][k]),  B operand = activations
//   (lane (j,h): feature k of row j),  C/D = 32 out-features x 32 rows.
// The C/D register map of this MFMA is  col = lane&31 (the row j),  row = sw_frow(reg, lane>>5)
// - the SAME lane holds row j before and after, so the accumulator registers of layer L are
// used, as they stand, as the B operands of layer L+1: register r of k-tile kt carries feature
// 32*kt + sw_frow(r,h), and the host-side pack kernel permutes W's columns to match
// (pack_kernels.hip).  No LDS round trip, no cross-lane movement between layers.
//
// Weights stream from L2 in "steps" of 1 KiB (4 MFMAs' A operands, one global_load_dwordx4
// per lane) through a ring of SW_RING steps kept in flight; the ring never drains between
// layers because the packed blob is laid out in execution order and ends with a copy of its
// own first SW_RING steps.  Biases (<= 18 KB per net) are copied to LDS once per workgroup.
// LDS per workgroup: bias 18 KiB + 4 waves x SW_RING KiB of weight ring (+ 28 KiB of resampling
// scratch in the render kernel).
//
// Reference arithmetic: model.py:39-62 (vallina_NeRF.forward), :273-296 (NeRFOriginal),
// :128-151 (DirectTemporalNeRF.query_time / forward).
#pragma once
#include <hip/hip_runtime.h>
#include "swnerf_common.h"
#include "lds_dma.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DSW_PROBE builds (tools/probe_segments.py; never the shipped library): shader-clock stamps around the parts of a
// tile, accumulated per wave in SGPRs and written, as raw 64-bit counters, into the ray's `weights` row.
#ifdef SW_PROBE
__device__ __forceinline__ unsigned long long sw_clock() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define SW_STAMP(var) const unsigned long long var = sw_clock()
#else
#define SW_STAMP(var)
#endif

// The weight ring lives in LDS and is filled by LDS-DMA (`global_load_lds_dwordx4`: global ->
// LDS with no VGPR destination): each wave owns SW_RING slots of 1 KiB and keeps SW_RING-1
// steps in flight.  Why not a register ring: hipcc (ROCm 7.2) sinks plain prefetch loads next
// to their use under this kernel's register pressure (load -> vmcnt(0) -> 4 MFMA, one exposed
// L2 round trip per step: 51 % of roofline), and inline-asm register loads are unsafe because
// the compiler copies/spills their destination VGPRs at phi merges while the data is still in
// flight (found by tools/isa_audit.py).  With LDS-DMA nothing the compiler can touch is ever
// pending: the DMA issue and the counted `s_waitcnt vmcnt(N)` are `asm volatile` with a memory
// clobber (so the compiler's own ds_read of a slot stays behind the wait that retires it),
// and the A operands reach the MFMAs through ordinary ds_read_b128 issued ONE STEP AHEAD, which
// the compiler tracks with lgkmcnt itself.  vmcnt retires in issue order and counts every VMEM
// op, so compiler-issued loads/stores in between only make a wait more conservative.
// Biases come from LDS too, so the hot loop's VMEM queue holds nothing but the DMA stream.
struct WStream {
    const char* base;        // wave-uniform: global byte address of the CURRENT step (SGPR pair)
    unsigned voff;           // lane * 16
    unsigned lds_addr;       // wave-uniform: LDS byte address of this wave's ring slot 0 (for M0)
    const float* ring;       // the same ring as a pointer, + lane*4 floats (for ds_read_b128)
    f32x4 a_cur;             // A operands of the CURRENT step (already read from LDS)
    const float* bias;       // LDS: this lane-half's 16 biases of the CURRENT output tile
};

template <int N>
__device__ __forceinline__ void ws_wait() {      // all but the N youngest VMEM ops have landed
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

__device__ __forceinline__ f32x4 ws_read(const WStream& ws, int slot) {
    return *reinterpret_cast<const f32x4*>(ws.ring + slot * SW_STEP_FLOATS);
}

template <int I>
__device__ __forceinline__ void ws_prime(WStream& ws) {
    if constexpr (I < SW_RING) {
        ws_dma(ws.base + I * 1024, ws.voff, ws.lds_addr + I * 1024);
        ws_prime<I + 1>(ws);
    }
}

// lds_ring: this wave's SW_RING KiB of LDS
__device__ __forceinline__ void ws_start(WStream& ws, const float* w, const float* lds_bias, float* lds_ring, int lane) {
    ws.base = reinterpret_cast<const char*>(w);
    ws.voff = (unsigned)lane * 16u;
    ws.lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_ring);
    ws.ring = lds_ring + lane * 4;
    ws.bias = lds_bias + (lane >> 5) * 16;
    ws_prime<0>(ws);
    ws_wait<SW_RING - 1>();
    ws.a_cur = ws_read(ws, 0);
}

// after the last segment of a stream the ring (and a_cur) already hold the blob's tail copy of
// the first SW_RING steps: rewind the pointers only.
__device__ __forceinline__ void ws_rewind(WStream& ws, const float* w, const float* lds_bias, int lane) {
    ws.base = reinterpret_cast<const char*>(w);
    ws.bias = lds_bias + (lane >> 5) * 16;
}

// Leave the stream the ring was following and continue at `w` (a stream whose head the ring does NOT hold): drain what is in
// flight, prime the ring afresh.  One exposed L2 round trip - for the kernels that evaluate one tile per wave (mlp_forward, the
// point query's entry into its views loop), never inside the fused passes' tile loop.
__device__ __forceinline__ void ws_restart(WStream& ws, const float* w) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ws.base = reinterpret_cast<const char*>(w);
    ws_prime<0>(ws);
    ws_wait<SW_RING - 1>();
    ws.a_cur = ws_read(ws, 0);
}

enum { SEG_ACC = 0, SEG_BIAS = 1, SEG_ZERO = 2, SEG_BIAS_SCALED = 3 };
// accumulator-init values of output tile n (bias tile n of the segment) for this lane half
template <int INIT>
__device__ __forceinline__ void seg_init_read(f32x16& b, const WStream& ws, int n) {
    if constexpr (INIT == SEG_BIAS || INIT == SEG_BIAS_SCALED) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ws.bias + n * SW_BIAS_TILE_FLOATS + 4 * g);
            b[4 * g + 0] = v[0]; b[4 * g + 1] = v[1]; b[4 * g + 2] = v[2]; b[4 * g + 3] = v[3];
        }
    }
}

// Side stores (training kernels): while a segment runs, its B operand `kin` - the previous layer's activations, or
// in the backward chain the gradient just masked - is written to a row-major [M, ld] buffer, ONE 16-byte store
// every few steps, right behind a step's weight DMA.  Issued as a burst at a layer boundary the same stores
// stall the wave twice: 32 KB per wave from all 1024 waves at once queue at HBM, and `vmcnt` retires in order, so
// the next counted wait sits behind every one of them.  Spread out they average ~1 TB/s and hide under the MFMAs.
// sp: this lane's row base (+ 4h); register r = 4g+e of k-tile n is feature 32n + 8g + 4h + e (16 contiguous bytes).
// mp/mv: one more 16-byte store at step 1 (the ReLU bit mask of the tile, see relu_bits).
struct SideStore {
    float* sp;
    float* mp; f32x4 mv;
};

template <int S, int NS, int NT, int KT, int INIT, int SK>
__device__ __forceinline__ void seg_steps(f32x16 (&out)[NT], const f32x16 (&kin)[KT], WStream& ws, f32x16& binit, float scale,
                                          const SideStore& ss) {
    if constexpr (S < NS) {
        constexpr int n = S / (KT * 4), kt = (S / 4) % KT, q = S % 4;
        constexpr int slot = S % SW_RING, nslot = (S + 1) % SW_RING;
        constexpr bool tile_first = (S % (KT * 4)) == 0, tile_last = ((S + 1) % (KT * 4)) == 0;
        if constexpr (tile_first) {
            // The init values were read from LDS one step ago (or by seg_mfma for tile 0): only ONE tile of
            // them is ever live.  Reading all NT tiles up front keeps 16*NT more accumulator registers busy
            // (the first MFMA of a tile takes them as srcC and writes a different dst), which is what pushed
            // the per-ray state of the render kernel into scratch.
            if constexpr (INIT == SEG_BIAS) out[n] = binit;
            else if constexpr (INIT == SEG_BIAS_SCALED) out[n] = binit * scale;
            else if constexpr (INIT == SEG_ZERO) {
#pragma unroll
                for (int r = 0; r < 16; ++r) out[n][r] = 0.f;
            }
        }
        // steps S+1 .. S+SW_RING-1 are in flight: retire the oldest and read it one step ahead
        ws_wait<SW_RING - 2>();
        const f32x4 a_next = ws_read(ws, nslot);
        if constexpr (tile_last && n + 1 < NT) seg_init_read<INIT>(binit, ws, n + 1);
        __builtin_amdgcn_sched_barrier(0);       // keep the ds_reads AHEAD of this step's MFMAs
        const f32x4 a = ws.a_cur;
        out[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], kin[kt][4 * q + 0], out[n], 0, 0, 0);
        out[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], kin[kt][4 * q + 1], out[n], 0, 0, 0);
        out[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], kin[kt][4 * q + 2], out[n], 0, 0, 0);
        out[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], kin[kt][4 * q + 3], out[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);       // ... and the refill BEHIND them (WAR on the slot)
        // refill the slot of step S (read during step S-1) with step S + SW_RING
        ws_dma(ws.base + (S + SW_RING) * 1024, ws.voff, ws.lds_addr + slot * 1024);
        if constexpr (SK > 0) {
            constexpr int total = 4 * SK, every = NS / total;
            static_assert(every >= 1, "segment too short for its side stores");
            if constexpr (S % every == every - 1 && S / every < total) {
                constexpr int idx = S / every, n2 = idx / 4, g = idx % 4;
                const f32x4 v = {kin[n2][4 * g], kin[n2][4 * g + 1], kin[n2][4 * g + 2], kin[n2][4 * g + 3]};
#ifndef TRAIN_EXP_NOSTORE
                *reinterpret_cast<f32x4*>(ss.sp + 32 * n2 + 8 * g) = v;
#endif
            }
            if constexpr (S == 1) {
                if (ss.mp) *reinterpret_cast<f32x4*>(ss.mp) = ss.mv;      // wave-uniform condition
            }
            __builtin_amdgcn_sched_barrier(0);   // the store stays HERE, not bunched up by the scheduler
        }
        ws.a_cur = a_next;
        seg_steps<S + 1, NS, NT, KT, INIT, SK>(out, kin, ws, binit, scale, ss);
    }
}

// out[n] (+)= sum_kt  Wtile(n,kt) . kin[kt]       NT output tiles, KT input tiles.
// INIT: how the accumulators start -
//   SEG_ACC (false) keep accumulating | SEG_BIAS (true) the bias tile (so no separate bias pass) |
//   SEG_ZERO zeros | SEG_BIAS_SCALED the bias tile times a per-lane scalar (backward: w_alpha * d sigma)
// SK > 0: the first SK k-tiles of `kin` are written out through `ss` while the segment runs (SideStore).
template <int NT, int KT, int INIT, int SK = 0>
__device__ __forceinline__ void seg_mfma(f32x16 (&out)[NT], const f32x16 (&kin)[KT], WStream& ws, float scale = 1.f,
                                         const SideStore& ss = SideStore{nullptr, nullptr, {0.f, 0.f, 0.f, 0.f}}) {
    constexpr int NS = NT * KT * 4;
    static_assert(NS % SW_RING == 0, "segment must keep the ring phase");
    f32x16 binit;
    seg_init_read<INIT>(binit, ws, 0);
    seg_steps<0, NS, NT, KT, INIT, SK>(out, kin, ws, binit, scale, ss);
    if (INIT == SEG_BIAS || INIT == SEG_BIAS_SCALED) ws.bias += NT * SW_BIAS_TILE_FLOATS;
    ws.base += NS * 1024;
}

// ---- ReLU bit masks (training) -----------------------------------------------------------------------
// One bit per activation of a 32-row x 256-feature tile: lane (j,h) packs its 8 x 16 values into 4 dwords, bit
// 16*(n&1) + r of dword n>>1 = (t[n][r] > 0).  64 lanes x 16 B = 1 KiB per (tile, layer): exactly one LDS-DMA
// step, which is how the backward chain fetches it (compiler-invisible, no exposed load latency) instead of
// re-reading 32 KiB of activations per layer just for their signs.
#define SW_MASK_LAYERS 9                          // h_0..h_7, views hidden (the deformation net uses the first 8)
#define SW_MASK_TILE_FLOATS (SW_MASK_LAYERS * 256)
template <int NT>
__device__ __forceinline__ f32x4 relu_bits(const f32x16 (&t)[NT]) {
    unsigned m[4] = {0u, 0u, 0u, 0u};
#ifdef TRAIN_EXP_NOBITS                                  // (timing experiment, WRONG gradients: tools/experiments/train/build.sh)
    return f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (int)bits clamped to [0,1]: 1 for positive floats, 0 for +0 and for -0 (sign bit = negative int)
            const int b = min(max(__float_as_int(t[n][r]), 0), 1);
            m[n >> 1] |= (unsigned)b << (16 * (n & 1) + r);
        }
    f32x4 v = {__uint_as_float(m[0]), __uint_as_float(m[1]), __uint_as_float(m[2]), __uint_as_float(m[3])};
    return v;
}

// t[n][r] = bit ? t[n][r] : 0
template <int NT>
__device__ __forceinline__ void mask_apply(const f32x4& mv, f32x16 (&t)[NT]) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = __float_as_int(mv[n >> 1]);
            const int keep = (m << (31 - (16 * (n & 1) + r))) >> 31;         // 0 or -1
            t[n][r] = __int_as_float(__float_as_int(t[n][r]) & keep);
        }
}

// ReLU as ONE v_max_f32: fmaxf(x, 0.f) costs two (hipcc first canonicalises x with v_max x,x because IEEE
// mode must quiet signalling NaNs; it folds a med3 clamp back into the same pair).  Plain VALU, so the
// statement needs no counters or wait states; not volatile, the scheduler may move it.
__device__ __forceinline__ float relu1(float x) {
    float y;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}

// ---- 1- and 3-output heads on the VALU ---------------------------------------------------------------
// res[o] = sum_f W[o][f] * x[f]  over the NT*32 features of a row.  Lane (j,h) holds the 16*NT features
// 32n + frow(r,h) of row j in x[n][r]; W[o] is stored like a bias vector (tile n: [h][r]), so each lane
// multiplies what it holds and the two halves of a row are added with one cross-half exchange.
// On return every lane (both halves) has the full sums.  Consumes NOUT*NT bias tiles.
template <int NOUT, int NT>
__device__ __forceinline__ void head_valu(const f32x16 (&x)[NT], WStream& ws, float (&res)[NOUT]) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(ws.bias + (o * NT + n) * SW_BIAS_TILE_FLOATS + 4 * g);
                acc = fmaf(w[0], x[n][4 * g + 0], acc); acc = fmaf(w[1], x[n][4 * g + 1], acc);
                acc = fmaf(w[2], x[n][4 * g + 2], acc); acc = fmaf(w[3], x[n][4 * g + 3], acc);
            }
        res[o] = acc + __shfl_xor(acc, 32, 64);
        // Pin the finished sum HERE (volatile asm keeps its order with the DMA statements of the next
        // segment).  Unpinned, the scheduler sinks half of the sigma dot product below the whole FEAT/VIEWS
        // tail - sigma is only consumed by the compositing - and keeps 64 weights plus the layer-7
        // activations alive across it, which is what spilled the per-ray state of the render kernel.
        asm volatile("" : "+v"(res[o]));
    }
    ws.bias += NOUT * NT * SW_BIAS_TILE_FLOATS;
}

// The same with a run-time output count (output_linear of a net without view directions: 4 or 5 channels):
// res[o] for o < nout, untouched beyond.  One loop body (256 FMAs per lane), not nout unrolled copies.
template <int NT>
__device__ __forceinline__ void head_valu_rt(const f32x16 (&x)[NT], WStream& ws, int nout, float (&res)[5]) {
#pragma nounroll
    for (int o = 0; o < nout; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(ws.bias + (o * NT + n) * SW_BIAS_TILE_FLOATS + 4 * g);
                acc = fmaf(w[0], x[n][4 * g + 0], acc); acc = fmaf(w[1], x[n][4 * g + 1], acc);
                acc = fmaf(w[2], x[n][4 * g + 2], acc); acc = fmaf(w[3], x[n][4 * g + 3], acc);
            }
        float v = acc + __shfl_xor(acc, 32, 64);
        asm volatile("" : "+v"(v));
        res[0] = o == 0 ? v : res[0]; res[1] = o == 1 ? v : res[1]; res[2] = o == 2 ? v : res[2];
        res[3] = o == 3 ? v : res[3]; res[4] = o == 4 ? v : res[4];
    }
    ws.bias += nout * NT * SW_BIAS_TILE_FLOATS;
}

// ---- activation tiles -> a row of a row-major [M, ld] buffer (training path) ---------------------
// register r = 4g+e of lane (j,h) of tile n is feature 32n + 8g + 4h + e of row j: 16 contiguous bytes.
// rowp = base + row*ld + 4h.  (Lanes past M use row M-1: they carry copies of that row, the stores are benign.)
template <int NT>
__device__ __forceinline__ void tiles_store(float* rowp, const f32x16 (&t)[NT]) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {t[n][4 * g], t[n][4 * g + 1], t[n][4 * g + 2], t[n][4 * g + 3]};
            *reinterpret_cast<f32x4*>(rowp + 32 * n + 8 * g) = v;
        }
}

// cooperative copy of a bias stream into LDS (whole block; ends with a barrier)
__device__ __forceinline__ void bias_to_lds(float* lds_bias, const float* gbias, int nfloats) {
    for (int i = threadIdx.x * 4; i < nfloats; i += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(lds_bias + i) = *reinterpret_cast<const f32x4*>(gbias + i);
    __syncthreads();
}

// ---- positional encodings into B-operand slots (slot maps: swnerf_common.h) --------------
__device__ __forceinline__ void pe_pos(float x0, float x1, float x2, int h, f32x16 (&e)[2]) {
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        float v;
        if (a < 30) {
            const int k = a / 3, c = a % 3;
            const float xc = (c == 0) ? x0 : ((c == 1) ? x1 : x2);
            v = sw_sin_or_cos(xc * (float)(1 << k), h);   // x * 2^k is exact (embedder.py:29,36)
        } else if (a == 30) {
            v = h ? x2 : x0;
        } else {
            v = h ? 0.f : x1;
        }
        e[a >> 4][a & 15] = v;
    }
}

__device__ __forceinline__ void pe_dir(float d0, float d1, float d2, int h, f32x16& e) {
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        float v = 0.f;
        if (a < 12) {
            const int k = a / 3, c = a % 3;
            const float dc = (c == 0) ? d0 : ((c == 1) ? d1 : d2);
            v = sw_sin_or_cos(dc * (float)(1 << k), h);
        } else if (a == 12) {
            v = h ? d2 : d0;
        } else if (a == 13) {
            v = h ? 0.f : d1;
        }
        e[a] = v;
    }
}

__device__ __forceinline__ void pe_time(float t, int h, f32x16& e) {
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        float v = 0.f;
        if (a < 10) v = sw_sin_or_cos(t * (float)(1 << a), h);
        else if (a == 10) v = h ? 0.f : t;
        e[a] = v;
    }
}

// The 2 embedding k-tiles are needed at layer 0 and again at the skip layer 5.  Keeping 32 registers alive
// across layers 1..4 tips this kernel into spilling (and a scratch reload inside a segment drains the DMA
// ring: scratch shares vmcnt), so they are parked in the wave's LDS slice [r][lane] in between.
#define SW_EMB_LDS_FLOATS (3 * 16 * 64)          // per wave: 2 position tiles + 1 view-direction tile
__device__ __forceinline__ void emb_park(float* lds_emb, int lane, const f32x16 (&e)[2]) {
#pragma unroll
    for (int a = 0; a < 32; ++a) lds_emb[a * 64 + lane] = e[a >> 4][a & 15];
}
__device__ __forceinline__ void emb_fetch(const float* lds_emb, int lane, f32x16 (&e)[2]) {
#pragma unroll
    for (int a = 0; a < 32; ++a) e[a >> 4][a & 15] = lds_emb[a * 64 + lane];
}

// the view-direction tile is a per-ray constant: left in registers the compiler hoists it out of the tile
// loop, keeps 16 registers alive for the whole ray and spills them into the VIEWS segment.
__device__ __forceinline__ void tile_park(float* lds_tile, int lane, const f32x16& e) {
#pragma unroll
    for (int a = 0; a < 16; ++a) lds_tile[a * 64 + lane] = e[a];
}
__device__ __forceinline__ void tile_fetch(const float* lds_tile, int lane, f32x16& e) {
#pragma unroll
    for (int a = 0; a < 16; ++a) e[a] = lds_tile[a * 64 + lane];
}

// ---- one trunk pass: 8 layers of width 256 with the skip at layer 5 ----------------------
// deform_pass: layer 0 also takes the time-embedding k-tile (model.py:129: cat[new_pts, t]).
// Returns with `in` = relu(layer-7 output) and head[0..2] = the head applied to it on every lane:
// alpha_linear (head[0] = sigma) for the canonical net, _time_out (dx) for the deformation net.
// TRAIN: every layer's post-ReLU activation h_l goes to act_row[256*l ...] (this lane's row of the row-major
// [M, SW_ACT_LD] buffer, + 4h) and its ReLU bit mask to mask_tile[256*l] (this lane's 16 bytes of the tile's
// SW_MASK_TILE_FLOATS), both as side stores of the NEXT segment, whose B operand h_l is.  h_7 has no next
// segment here: store_last writes it on the spot, otherwise the caller side-stores it (in `in`, mask in *mb).
// XS (fused training pass): the two position-encoding k-tiles go to xs_row (+ 4h; slot order, sw_xs_col) as side
// stores of layer 0, whose B operand they are.
// NOHEAD: stop after layer 7 (`in` = relu(h7)); the caller applies its own head (output_linear of a net without view
// directions, head_valu_rt).
// TB (D-NeRF, fused passes): layer 0 of the deformation pass starts from the per-ray TIME tile lds_tb (see below).
template <bool DNERF, bool TRAIN = false, bool XS = false, bool NOHEAD = false, bool TB = false>
__device__ __forceinline__ void trunk_pass(const f32x16 (&emb)[2], float* lds_emb, float t, bool deform_pass, int h,
                                           f32x16 (&in)[8], f32x16 (&out)[8], float (&head)[3], WStream& ws,
                                           float* act_row = nullptr, float* mask_tile = nullptr, bool store_last = false,
                                           f32x4* mb = nullptr, float* xs_row = nullptr, const float* lds_tb = nullptr) {
    const int lane_ = threadIdx.x & 63;
    emb_park(lds_emb, lane_, emb);
    f32x4 mbits = {0.f, 0.f, 0.f, 0.f};
#pragma nounroll
    for (int l = 0; l < 8; ++l) {
        if (l == 0) {
            if (DNERF && deform_pass) {
                // _time.0 on cat[gamma(x), gamma(t)] (model.py:129) as TWO segments of the stream: TIME (bias + the gamma(t) columns),
                // then the gamma(x) columns on top.  TB (the fused passes: ONE time per ray): the wave has evaluated TIME
                // once (time_bias_tile) and this tile's accumulators start from that per-ray tile - 128 MFMAs per tile less; otherwise (time
                // per ROW: mlp_forward, the training forward of the op path): both segments here.  The same sequence of additions
                // either way, so the two paths give the same bits.
                if constexpr (TB) {
                    const float* keep = ws.bias;
                    ws.bias = lds_tb + (lane_ & 32 ? 16 : 0);
                    if (XS) {
                        seg_mfma<8, 2, SEG_BIAS, 2>(out, emb, ws, 1.f, SideStore{xs_row, nullptr, mbits});   // gamma(x) -> xs ...
                        f32x16 tt;
                        pe_time(t, h, tt);                                                                    // ... and gamma(t), on the spot
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = {tt[4 * g], tt[4 * g + 1], tt[4 * g + 2], tt[4 * g + 3]};
                            *reinterpret_cast<f32x4*>(xs_row + 64 + 8 * g) = v;
                        }
                    } else {
                        seg_mfma<8, 2, SEG_BIAS>(out, emb, ws);
                    }
                    ws.bias = keep;
                } else {
                    f32x16 k1[1];
                    pe_time(t, h, k1[0]);
                    seg_mfma<8, 1, SEG_BIAS>(out, k1, ws);
                    seg_mfma<8, 2, SEG_ACC>(out, emb, ws);
                }
            } else if (XS) {
                seg_mfma<8, 2, SEG_BIAS, 2>(out, emb, ws, 1.f, SideStore{xs_row, nullptr, mbits});
            } else {
                seg_mfma<8, 2, SEG_BIAS>(out, emb, ws);
            }
        } else {
            if (TRAIN) {
                const SideStore ss{act_row + 256 * (l - 1), mask_tile + 256 * (l - 1), mbits};
                seg_mfma<8, 8, SEG_BIAS, TRAIN ? 8 : 0>(out, in, ws, 1.f, ss);
            } else {
                seg_mfma<8, 8, SEG_BIAS>(out, in, ws);
            }
            if (l == 5) {                                        // skip: cat[input_pts, h] (model.py:45-46)
                f32x16 e2[2];
                emb_fetch(lds_emb, lane_, e2);
                seg_mfma<8, 2, SEG_ACC>(out, e2, ws);
            }
        }
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) in[n][r] = relu1(out[n][r]);
        if (TRAIN) mbits = relu_bits<8>(in);
    }
    if (TRAIN) {
        if (store_last) {
            tiles_store<8>(act_row + 256 * 7, in);
            *reinterpret_cast<f32x4*>(mask_tile + 256 * 7) = mbits;
        } else {
            *mb = mbits;
        }
    }
    if constexpr (NOHEAD) return;
    // head biases: one tile right behind the weight tiles, the same 16 floats in both lane halves:
    // [b_alpha, b_r, b_g, b_b] (canonical) / [b_dx0, b_dx1, b_dx2] (deformation)
    if (DNERF && deform_pass) {
        head_valu<3, 8>(in, ws, head);
        head[0] += ws.bias[0]; head[1] += ws.bias[1]; head[2] += ws.bias[2];
    } else {
        float s1[1];
        head_valu<1, 8>(in, ws, s1);
        head[0] = s1[0] + ws.bias[0]; head[1] = 0.f; head[2] = 0.f;
    }
    ws.bias += SW_BIAS_TILE_FLOATS;
}

// ---- canonical tail: [feature_linear folded into] views_linears[0] + relu -> rgb_linear ---
// `in` = relu(layer 7).  On return rgb[0..2] = raw rgb of row j on every lane (model.py:49-58).
// feature_linear has no activation, so the packed stream carries W_vf = Wv[:, :256] . W_f and b_vf = Wv[:, :256] . b_f + b_v
// (swnerf_common.h SW_CANON_STEPS, pack_kernels.hip fold_views_kernel).
// hb_rgb: the head-bias tile (LDS) saved by the caller: [b_alpha, b_r, b_g, b_b].
//
// Two forms of the view layer:
//  * canon_tail_rows: directions vary per ROW (mlp_forward on embedded rows, the point query): one 4 x 9 segment on
//    [gamma(d) | h7] taken from the blob's views-loop stream (ws_restart), initialised from the b_vf tiles;
//  * canon_tail / canon_tail_train: the fused passes, ONE direction per ray: the wave has evaluated
//    c = Wv[:, 256:] gamma(d) + b_vf once (view_bias_tile below: 64 MFMAs per RAY) and every tile runs a 4 x 8 segment on h7
//    whose accumulators start from c - 64 MFMAs per TILE less.
#define SW_VB_LDS_FLOATS (4 * SW_BIAS_TILE_FLOATS)       // per wave: the per-ray init tiles of the view layer, [n][h][r]

// once per ray, right after ws_start on a stream that begins with the DIR prefix: vb[n][h][r] = c[32n + frow(r,h)]
__device__ __forceinline__ void view_bias_tile(const f32x16& demb, float* lds_vb, int lane, WStream& ws) {
    f32x16 k1[1], c[4];
    k1[0] = demb;
    seg_mfma<4, 1, SEG_BIAS>(c, k1, ws);                     // every column j of the result is the same: gamma(d) is the ray's
    if ((lane & 31) == 0) {
        float* o = lds_vb + (lane >> 5) * 16;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {c[n][4 * g], c[n][4 * g + 1], c[n][4 * g + 2], c[n][4 * g + 3]};
                *reinterpret_cast<f32x4*>(o + n * SW_BIAS_TILE_FLOATS + 4 * g) = v;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The same for the deformation net's layer 0 (D-NeRF fused passes, right behind view_bias_tile: the stream's TIME segment follows
// its DIR prefix): tb[n][h][r] = (_time.0.bias + _time.0.weight[:, Cpos:] gamma(t))[32n + frow(r,h)], 8 tiles, once per ray.
#define SW_TB_LDS_FLOATS (8 * SW_BIAS_TILE_FLOATS)
__device__ __forceinline__ void time_bias_tile(float t, int h, float* lds_tb, int lane, WStream& ws) {
    f32x16 k1[1], c[8];
    pe_time(t, h, k1[0]);
    seg_mfma<8, 1, SEG_BIAS>(c, k1, ws);
    if ((lane & 31) == 0) {
        float* o = lds_tb + (lane >> 5) * 16;
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {c[n][4 * g], c[n][4 * g + 1], c[n][4 * g + 2], c[n][4 * g + 3]};
                *reinterpret_cast<f32x4*>(o + n * SW_BIAS_TILE_FLOATS + 4 * g) = v;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void canon_tail(const f32x16 (&in)[8], const float* lds_vb, float (&rgb)[3], const float* hb_rgb, WStream& ws) {
    f32x16 hv[4];
    const float* keep = ws.bias;
    ws.bias = lds_vb + (threadIdx.x & 32 ? 16 : 0);
    seg_mfma<4, 8, SEG_BIAS>(hv, in, ws);
    ws.bias = keep;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[n][r] = relu1(hv[n][r]);
    head_valu<3, 4>(hv, ws, rgb);
    rgb[0] += hb_rgb[1]; rgb[1] += hb_rgb[2]; rgb[2] += hb_rgb[3];
}

// The same tail in the fused training passes: h7 (`in`, ReLU mask `mb`) is side-stored by the view layer's segment, whose B
// operand it is; the view hidden layer and its mask are stored on the spot (act_row / mask_tile: trunk_pass).  `feature` does
// not exist: the weight gradients on both sides of feature_linear follow from G = d pre_hv^T . h7 (swnerf_feature_finish).
__device__ __forceinline__ void canon_tail_train(const f32x16 (&in)[8], const float* lds_vb, float (&rgb)[3], const float* hb_rgb,
                                                 WStream& ws, float* act_row, float* mask_tile, const f32x4& mb) {
    f32x16 hv[4];
    const float* keep = ws.bias;
    ws.bias = lds_vb + (threadIdx.x & 32 ? 16 : 0);
    seg_mfma<4, 8, SEG_BIAS, 8>(hv, in, ws, 1.f, SideStore{act_row + 256 * 7, mask_tile + 256 * 7, mb});
    ws.bias = keep;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[n][r] = relu1(hv[n][r]);
#ifndef TRAIN_EXP_NOBURST                               // (timing experiment: tools/experiments/train/build.sh)
    tiles_store<4>(act_row + SW_ACT_HV, hv);
#endif
    *reinterpret_cast<f32x4*>(mask_tile + 256 * 8) = relu_bits<4>(hv);
    head_valu<3, 4>(hv, ws, rgb);
    rgb[0] += hb_rgb[1]; rgb[1] += hb_rgb[2]; rgb[2] += hb_rgb[3];
}

// Per-row directions: the 4 x 9 view layer from the views-loop stream, whose k-tile order is [gamma(d) | h7] - the accumulators
// take b_vf, then the direction terms, then the h7 terms: EXACTLY the sequence of view_bias_tile + canon_tail, so the op path and
// the fused pass produce the same bits (the resampling downstream is a discontinuous function of the coarse weights: two paths of
// one library must not differ in the last bit there).  The caller has put the ring onto `wvl` (ws_restart, or a
// previous turn whose tail is the stream's own head) and ws.base = wvl; bvf: the b_vf tiles (LDS, + 16 h).
__device__ __forceinline__ void canon_tail_rows(const f32x16 (&in)[8], const f32x16& demb, f32x16 (&hv)[4], const float* bvf, WStream& ws) {
    f32x16 k9[9];
    k9[0] = demb;                                           // cat[h7, input_views], direction k-tile first
#pragma unroll
    for (int n = 0; n < 8; ++n) k9[1 + n] = in[n];
    const float* keep = ws.bias;
    ws.bias = bvf;
    seg_mfma<4, 9, SEG_BIAS>(hv, k9, ws);
    ws.bias = keep;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[n][r] = relu1(hv[n][r]);
}
