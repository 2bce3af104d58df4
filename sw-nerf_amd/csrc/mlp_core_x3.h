// mlp_core_x3.h - the 8x256 NeRF MLP on the bf16 matrix pipe at (nearly) fp32 accuracy: "bf16x3".
//
// Every fp32 operand is split into two bf16 halves, v = hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 significant
// bits together), and a product W.x is evaluated as three v_mfma_f32_32x32x16_bf16 with fp32 accumulation,
//     W_hi.x_hi + W_hi.x_lo + W_lo.x_hi          (the dropped W_lo.x_lo term is 2^-18 relative),
// which is 3/16 of the cycles of the same contraction on v_mfma_f32_32x32x2_f32.  Same orientation as mlp_core.h
// (A = weights, B = activations, one wave = 32 rows, lane (j,h) keeps row j throughout), and again the C/D register map
// makes a layer's accumulators the next layer's B operand without leaving the lane: k-block 2n+c of a 256-wide input is
// registers 8c..8c+7 of accumulator tile n, i.e. features 32n + sw_frow(8c+p, h) for position p of lane half h, and the
// pack kernel orders the weight columns to match (x3_kernels.hip).  An opt-in path: it is NOT the parity path and not
// what bench.py's headline measures (fp32); tolerances and PSNR are recorded where it is tested.
//
// Weight stream.  At this MFMA rate a per-wave private stream (mlp_core.h) would need 4 x 2 KiB per 96 cycles from L2
// into every CU - about 50 TB/s chip-wide - so the four waves of a workgroup SHARE one ring in LDS and run in step:
// a chunk is 8 groups of [A_hi 1 KiB][A_lo 1 KiB] (one k-block for all 8 output tiles), the ring holds X3_NSLOT
// chunks, every wave issues a quarter of each chunk by LDS-DMA and all of them read all of it.  One s_barrier per
// chunk (768 MFMA cycles) both publishes the next chunk (each wave has waited for its own quarter with a counted
// vmcnt before arriving) and frees the slot of the previous one for the DMA issued right behind the barrier.
//
// Reference arithmetic: model.py:39-62 (vallina_NeRF.forward), :273-296 (NeRFOriginal).
#pragma once
#include <hip/hip_runtime.h>
#include "swnerf_common.h"
#include "lds_dma.h"
#include "mlp_core.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef X3_AHEAD
#define X3_AHEAD 2        // A operands are read from LDS this many groups ahead of their MFMAs
#endif
#define X3_NSLOT 6
#define X3_GROUP_BYTES 2048
#define X3_CHUNK_GROUPS 8
#define X3_CHUNK_BYTES (X3_CHUNK_GROUPS * X3_GROUP_BYTES)
#define X3_RING_FLOATS (X3_NSLOT * X3_CHUNK_BYTES / 4)

struct XStream {
    const char* gnext;       // wave-uniform: global address of this wave's quarter of the next chunk to ISSUE
    unsigned voff;           // lane * 16
    unsigned lds_q;          // wave-uniform: LDS byte address of this wave's quarter of ring slot 0
    unsigned islot;          // wave-uniform: ring slot the next issued chunk goes to
    unsigned idst;           // wave-uniform: LDS byte address of this wave's quarter of the slot being filled
    unsigned rslot;          // wave-uniform: ring slot of the chunk being READ
    const char* ring_lane;   // LDS: ring + lane * 16
    const char* rd;          // LDS: ring_lane + rslot * X3_CHUNK_BYTES
    const char* rd_next;     // ... of the chunk after it
    u32x4 ahi, alo;          // A operands of the CURRENT group (already read from LDS)
    u32x4 a2hi, a2lo;        // ... of the group after it (X3_AHEAD == 2)
    const float* bias;       // LDS: this lane half's 16 accumulator-init values of the current output tile
#ifdef SW_PROBE
    unsigned long long pc[3];   // cycles in: MFMA segments | accumulator -> (hi, lo) splits + heads | gamma(x) evaluations
#endif
};
#ifdef SW_PROBE
#define X3_PROBE(i, t0) xs.pc[i] += sw_clock() - t0
#else
#define X3_PROBE(i, t0)
#endif

// One chunk = 4 DMA instructions per wave.  Back to back they stall the wave while the matrix pipe runs dry (each
// waits for the address path), so in the steady state they go out ONE PER GROUP, each behind a group's MFMAs
// (x3_issue_part); only the priming chunks are issued in one go.
// wave-uniform by construction; said explicitly, because in the two-net kernel the compiler's divergence analysis gives
// up on the pointer and hands the "s" operand of the DMA statement a VGPR pair
__device__ __forceinline__ const char* x3_uniform(const char* p) {
    const unsigned long long v = (unsigned long long)(size_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>((size_t)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ void x3_issue_part(XStream& xs, int i) {
#ifndef X3_EXP_NODMA                             // timing experiment: only the priming chunks are ever loaded
    ws_dma(x3_uniform(xs.gnext) + i * 1024, xs.voff, __builtin_amdgcn_readfirstlane(xs.idst) + i * 1024);
#endif
    if (i == 3) xs.gnext += X3_CHUNK_BYTES;
}
__device__ __forceinline__ void x3_issue_begin(XStream& xs) {
    xs.idst = xs.lds_q + xs.islot * X3_CHUNK_BYTES;
    xs.islot = (xs.islot + 1 == X3_NSLOT) ? 0u : xs.islot + 1;
}
__device__ __forceinline__ void x3_issue(XStream& xs) {
    x3_issue_begin(xs);
#pragma unroll
    for (int i = 0; i < 4; ++i) ws_dma(x3_uniform(xs.gnext) + i * 1024, xs.voff, __builtin_amdgcn_readfirstlane(xs.idst) + i * 1024);
    xs.gnext += X3_CHUNK_BYTES;
}

// the workgroup barrier alone: no fence, no drain of the DMA queue (which __syncthreads() would add)
__device__ __forceinline__ void x3_barrier() {
    __builtin_amdgcn_sched_barrier(0);
#ifndef X3_EXP_NOBARRIER                         // timing experiments only (tools/experiments/x3): results are then garbage
    asm volatile("s_barrier" ::: "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void x3_read(const char* p, u32x4& hi, u32x4& lo) {
    hi = *reinterpret_cast<const u32x4*>(p);
    lo = *reinterpret_cast<const u32x4*>(p + 1024);
}

// Called by ALL four waves (the ring is shared).  w: the x3 weight stream; lds_ring: the workgroup's ring.
__device__ __forceinline__ void x3_start(XStream& xs, const char* w, const float* lds_bias, const float* lds_ring, int lane, int wv) {
    xs.gnext = w + wv * 4096;
    xs.voff = (unsigned)lane * 16u;
    xs.lds_q = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_ring) + (unsigned)wv * 4096u;
    xs.islot = 0; xs.rslot = 0;
    xs.ring_lane = reinterpret_cast<const char*>(lds_ring) + lane * 16;
    xs.rd = xs.ring_lane;
    xs.rd_next = xs.ring_lane;
    xs.bias = lds_bias + (lane >> 5) * 16;
#ifdef SW_PROBE
    xs.pc[0] = xs.pc[1] = xs.pc[2] = 0ull;
#endif
    // the steady state (x3_groups) begins the refill round of chunk q + X3_NSLOT - 1 at group 7 - X3_AHEAD of chunk q and
    // issues its 4 parts behind 4 consecutive groups, so a round straddles the chunk boundary: enter that state as if
    // chunk -1 had just begun the round of chunk X3_NSLOT - 2 and issued the parts that precede group 0
#pragma unroll
    for (int c = 0; c < X3_NSLOT - 2; ++c) x3_issue(xs);
    x3_issue_begin(xs);
#pragma unroll
    for (int i = 0; i <= X3_AHEAD; ++i) x3_issue_part(xs, i);
    ws_wait<4 * (X3_NSLOT - 3) + X3_AHEAD + 1>();   // this wave's quarter of chunk 0
    x3_barrier();                                   // ... and everybody else's
    x3_read(xs.rd, xs.ahi, xs.alo);
    x3_read(xs.rd + X3_GROUP_BYTES, xs.a2hi, xs.a2lo);
}

// At group 6 of the chunk being read: publish the next chunk, free the previous one's slot and refill it.
__device__ __forceinline__ void x3_advance(XStream& xs) {
#ifndef X3_EXP_NOWAIT
    ws_wait<4 * (X3_NSLOT - 3)>();           // own quarter of the NEXT chunk has landed (the X3_NSLOT-3 behind it may fly)
#endif
    x3_barrier();
    x3_issue_begin(xs);                      // the slot of the PREVIOUS chunk: every wave is past it (parts: x3_groups)
    const unsigned ns = (xs.rslot + 1 == X3_NSLOT) ? 0u : xs.rslot + 1;
    xs.rslot = ns;
    xs.rd_next = xs.ring_lane + ns * X3_CHUNK_BYTES;
}

// after the last segment of a tile the ring already holds (or has in flight) the stream's tail = a copy of its
// first chunks: rewind the issue pointer by the net's chunks
__device__ __forceinline__ void x3_rewind(XStream& xs, int net_chunks, const float* lds_bias, int lane) {
    xs.gnext -= (size_t)net_chunks * X3_CHUNK_BYTES;
    xs.bias = lds_bias + (lane >> 5) * 16;
}

__device__ __forceinline__ unsigned x3_cvt_pk(float a, float b) {       // [bf16(a) | bf16(b) << 16], round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// 16 fp32 values of a lane (one accumulator / embedding tile) -> two k-blocks of (hi, lo) B operands
// `floor`: 0 for a ReLU layer, -inf for none (one v_max_f32 either way; see relu1 for why it is spelled in asm)
__device__ __forceinline__ float x3_floor(float x, float floor) {
    float y;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(y) : "v"(floor), "v"(x));
    return y;
}
template <bool FLOOR>
__device__ __forceinline__ void x3_split(const f32x16& v, u32x4& hi0, u32x4& lo0, u32x4& hi1, u32x4& lo1, float floor = 0.f) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float a = FLOOR ? x3_floor(v[2 * q], floor) : v[2 * q], b = FLOOR ? x3_floor(v[2 * q + 1], floor) : v[2 * q + 1];
        const unsigned h2 = x3_cvt_pk(a, b);
        const float ra = a - __uint_as_float(h2 << 16), rb = b - __uint_as_float(h2 & 0xffff0000u);
        const unsigned l2 = x3_cvt_pk(ra, rb);
        if (q < 4) { hi0[q] = h2; lo0[q] = l2; } else { hi1[q - 4] = h2; lo1[q - 4] = l2; }
    }
}

template <int TERMS>
__device__ __forceinline__ f32x16 x3_mfma(const u32x4& ahi, const u32x4& alo, const u32x4& bhi, const u32x4& blo, f32x16 acc) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, ahi), bh = __builtin_bit_cast(bf16x8, bhi);
    if constexpr (TERMS == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, alo), bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, blo), acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

template <int G, int NG, int NT, int KB0, int TERMS, int NB>
__device__ __forceinline__ void x3_groups(f32x16 (&acc)[NT], const u32x4 (&bhi)[NB], const u32x4 (&blo)[NB], XStream& xs) {
    if constexpr (G < NG) {
        constexpr int g = G % X3_CHUNK_GROUPS, kb = KB0 + G / NT, n = G % NT;
        if constexpr (g == X3_CHUNK_GROUPS - 1 - X3_AHEAD) x3_advance(xs);
        u32x4 nhi, nlo;                                         // the operands of group G + X3_AHEAD
        if constexpr (g + X3_AHEAD >= X3_CHUNK_GROUPS) x3_read(xs.rd_next + (g + X3_AHEAD - X3_CHUNK_GROUPS) * X3_GROUP_BYTES, nhi, nlo);
        else x3_read(xs.rd + (g + X3_AHEAD) * X3_GROUP_BYTES, nhi, nlo);
        __builtin_amdgcn_sched_barrier(0);                      // keep the ds_reads AHEAD of this group's MFMAs
        acc[n] = x3_mfma<TERMS>(xs.ahi, xs.alo, bhi[kb], blo[kb], acc[n]);
        __builtin_amdgcn_sched_barrier(0);
        {   // the refill of the slot freed at this chunk's barrier: parts 0..3 behind the MFMAs of 4 consecutive groups
            constexpr int part = (g - (X3_CHUNK_GROUPS - 1 - X3_AHEAD) + X3_CHUNK_GROUPS) % X3_CHUNK_GROUPS;
            if constexpr (part < 4) {
                x3_issue_part(xs, part);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (g == X3_CHUNK_GROUPS - 1) xs.rd = xs.rd_next;
#if X3_AHEAD == 2
        xs.ahi = xs.a2hi; xs.alo = xs.a2lo; xs.a2hi = nhi; xs.a2lo = nlo;
#else
        xs.ahi = nhi; xs.alo = nlo;
#endif
        x3_groups<G + 1, NG, NT, KB0, TERMS, NB>(acc, bhi, blo, xs);
    }
}

// acc[n] (+)= sum over k-blocks KB0 .. KB0+KB-1 of  Wblock(n,kb) . B[kb]      (k-block outer, output tile inner)
// INIT: SEG_BIAS (accumulators start from the next NT bias tiles) or SEG_ACC.
template <int NT, int KB0, int KB, int INIT, int TERMS, int NB>
__device__ __forceinline__ void x3_seg(f32x16 (&acc)[NT], const u32x4 (&bhi)[NB], const u32x4 (&blo)[NB], XStream& xs) {
    static_assert((NT * KB) % X3_CHUNK_GROUPS == 0, "a segment is a whole number of chunks");
    static_assert(KB0 + KB <= NB, "k-blocks out of range");
    if constexpr (INIT == SEG_BIAS) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xs.bias + n * SW_BIAS_TILE_FLOATS + 4 * q);
                acc[n][4 * q + 0] = v[0]; acc[n][4 * q + 1] = v[1]; acc[n][4 * q + 2] = v[2]; acc[n][4 * q + 3] = v[3];
                if (q == 3) __builtin_amdgcn_sched_barrier(0);   // tile by tile: 128 values in flight at once is 128 more VGPRs
            }
        xs.bias += NT * SW_BIAS_TILE_FLOATS;
    }
    x3_groups<0, NT * KB, NT, KB0, TERMS, NB>(acc, bhi, blo, xs);
}

// one output head on the VALU over fp32 values (the accumulators before they are split): partial dot product of this
// lane's 16 features of tile n with bias-style weight tile `wt`
__device__ __forceinline__ float x3_head_part(const f32x16& x, const float* wt, float acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wt + 4 * g);
        acc = fmaf(w[0], x[4 * g + 0], acc); acc = fmaf(w[1], x[4 * g + 1], acc);
        acc = fmaf(w[2], x[4 * g + 2], acc); acc = fmaf(w[3], x[4 * g + 3], acc);
    }
    return acc;
}

// The canonical net on one 32-row tile at positions (px,py,pz); lds_dir: the parked gamma(d) tile of the ray.
// On return sigma / rgb[3] = the raw outputs of row j on every lane (model.py:49-58).
// Register budget (one wave per SIMD, 256 VGPR + 256 AGPR): 128 accumulators + 128 registers of (hi, lo) activations
// are the floor; gamma(x) is therefore not kept for the skip layer but evaluated again in front of it (3 % of a
// tile's VALU work, no LDS), and gamma(d) is fetched just before the view layer.
template <int TERMS>
__device__ __forceinline__ void x3_canon(float px, float py, float pz, int h, const float* lds_dir, int lane,
                                         float& sigma, float (&rgb)[3], XStream& xs) {
    u32x4 bhi[16], blo[16];
    f32x16 acc[8];
    {   // pts_linears[0] on gamma(x)
        SW_STAMP(q1);
        f32x16 emb[2];
        pe_pos(px, py, pz, h, emb);
        u32x4 ehi[4], elo[4];
        x3_split<false>(emb[0], ehi[0], elo[0], ehi[1], elo[1]);
        x3_split<false>(emb[1], ehi[2], elo[2], ehi[3], elo[3]);
        X3_PROBE(2, q1);
        SW_STAMP(q2);
        x3_seg<8, 0, 4, SEG_BIAS, TERMS>(acc, ehi, elo, xs);
        X3_PROBE(0, q2);
        SW_STAMP(q3);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            x3_split<true>(acc[n], bhi[2 * n], blo[2 * n], bhi[2 * n + 1], blo[2 * n + 1], 0.f);
            __builtin_amdgcn_sched_barrier(0);                  // tile by tile (register pressure)
        }
        X3_PROBE(1, q3);
    }
    const float* hb = nullptr;
    // pts_linears[1..7] share ONE body: the accumulators then have one home in the register file (three differently shaped bodies
    // in one loop made the compiler shuffle 48 of them through scratch).  feature_linear is folded into the view layer (round 4,
    // swnerf_common.h SW_CANON_STEPS): the view layer below runs on the split of relu(h_7) with the folded weights.
#pragma nounroll
    for (int l = 1; l <= 7; ++l) {
        SW_STAMP(q0);
        x3_seg<8, 0, 16, SEG_BIAS, TERMS>(acc, bhi, blo, xs);
        X3_PROBE(0, q0);
        if (l == 5) {                                               // cat[input_pts, h] (model.py:45-46): ... then gamma(x),
            SW_STAMP(q1);                                           // evaluated again HERE: hoisted it would sit in scratch
            f32x16 emb[2];                                          // across layers 1..4 and a scratch reload drains the DMA ring
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
            pe_pos(px, py, pz, h, emb);
            u32x4 ehi[4], elo[4];
            x3_split<false>(emb[0], ehi[0], elo[0], ehi[1], elo[1]);
            x3_split<false>(emb[1], ehi[2], elo[2], ehi[3], elo[3]);
            X3_PROBE(2, q1);
            SW_STAMP(q2);
            x3_seg<8, 0, 4, SEG_ACC, TERMS>(acc, ehi, elo, xs);
            X3_PROBE(0, q2);
        }
        SW_STAMP(q3);
        if (l == 7) {
            // alpha_linear on relu(h_7) in fp32, before the split: 8 weight tiles, then the head-bias tile
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = relu1(acc[n][r]);
                s = x3_head_part(t, xs.bias + n * SW_BIAS_TILE_FLOATS, s);
                __builtin_amdgcn_sched_barrier(0);
            }
            s += __shfl_xor(s, 32, 64);
            sigma = s + xs.bias[8 * SW_BIAS_TILE_FLOATS];
            hb = xs.bias + 8 * SW_BIAS_TILE_FLOATS;                 // [b_alpha, b_r, b_g, b_b]
            xs.bias += 9 * SW_BIAS_TILE_FLOATS;
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            x3_split<true>(acc[n], bhi[2 * n], blo[2 * n], bhi[2 * n + 1], blo[2 * n + 1], 0.f);
            __builtin_amdgcn_sched_barrier(0);
        }
        X3_PROBE(1, q3);
    }
    f32x16 hv[4];
    SW_STAMP(q6);
    x3_seg<4, 0, 16, SEG_BIAS, TERMS>(hv, bhi, blo, xs);           // views_linears[0] . feature_linear (folded) on relu(h_7) ...
    X3_PROBE(0, q6);
    {
        f32x16 demb;
        tile_fetch(lds_dir, lane, demb);
        u32x4 dhi[2], dlo[2];
        x3_split<false>(demb, dhi[0], dlo[0], dhi[1], dlo[1]);
        SW_STAMP(q7);
        x3_seg<4, 0, 2, SEG_ACC, TERMS>(hv, dhi, dlo, xs);         // ... then gamma(d)
        X3_PROBE(0, q7);
    }
    SW_STAMP(q8);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = relu1(hv[n][r]);
            s = x3_head_part(t, xs.bias + (o * 4 + n) * SW_BIAS_TILE_FLOATS, s);
        }
        rgb[o] = s + __shfl_xor(s, 32, 64) + hb[1 + o];
    }
    X3_PROBE(1, q8);
}

// DirectTemporalNeRF (model.py:128-151) on one 32-row tile: ONE body for both nets, like the fp32 kernel - `deform`
// selects the deformation net (layer 0 also takes gamma(t); 7 more layers; head = _time_out, 3 outputs in head[]) or the
// canonical net (as x3_canon; head[0] = sigma, rgb[]).  gamma(d) is evaluated here from the view direction (no LDS tile:
// the two bias-tile sets need its room).
template <int TERMS>
__device__ __forceinline__ void x3_net_dn(float px, float py, float pz, float ft, bool deform, int h, float v0, float v1, float v2,
                                          float (&head)[3], float (&rgb)[3], XStream& xs) {
    u32x4 bhi[16], blo[16];
    f32x16 acc[8];
    {
        f32x16 emb[2];
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
        pe_pos(px, py, pz, h, emb);
        u32x4 ehi[6], elo[6];
        x3_split<false>(emb[0], ehi[0], elo[0], ehi[1], elo[1]);
        x3_split<false>(emb[1], ehi[2], elo[2], ehi[3], elo[3]);
        if (deform) {                                               // cat[new_pts, t] (model.py:129)
            f32x16 te;
            pe_time(ft, h, te);
            x3_split<false>(te, ehi[4], elo[4], ehi[5], elo[5]);
            x3_seg<8, 0, 6, SEG_BIAS, TERMS>(acc, ehi, elo, xs);
        } else {
            x3_seg<8, 0, 4, SEG_BIAS, TERMS>(acc, ehi, elo, xs);
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            x3_split<true>(acc[n], bhi[2 * n], blo[2 * n], bhi[2 * n + 1], blo[2 * n + 1], 0.f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float* hb = nullptr;
#pragma nounroll
    for (int l = 1; l <= 7; ++l) {                                  // (feature_linear is folded into the canonical net's view layer)
        x3_seg<8, 0, 16, SEG_BIAS, TERMS>(acc, bhi, blo, xs);
        if (l == 5) {
            f32x16 emb[2];
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
            pe_pos(px, py, pz, h, emb);
            u32x4 ehi[4], elo[4];
            x3_split<false>(emb[0], ehi[0], elo[0], ehi[1], elo[1]);
            x3_split<false>(emb[1], ehi[2], elo[2], ehi[3], elo[3]);
            x3_seg<8, 0, 4, SEG_ACC, TERMS>(acc, ehi, elo, xs);
        }
        if (l == 7) {
            // the head on relu(h_7) in fp32: alpha_linear (1 output) or _time_out (3), weight tiles then the head-bias tile
            const int nout = deform ? 3 : 1;
#pragma nounroll
            for (int o = 0; o < nout; ++o) {
                float s = 0.f;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    f32x16 t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = relu1(acc[n][r]);
                    s = x3_head_part(t, xs.bias + (o * 8 + n) * SW_BIAS_TILE_FLOATS, s);
                    __builtin_amdgcn_sched_barrier(0);
                }
                s += __shfl_xor(s, 32, 64);
                const float v = s + xs.bias[nout * 8 * SW_BIAS_TILE_FLOATS + o];
                if (o == 0) head[0] = v; else if (o == 1) head[1] = v; else head[2] = v;
            }
            hb = xs.bias + nout * 8 * SW_BIAS_TILE_FLOATS;           // canonical: [b_alpha, b_r, b_g, b_b]
            xs.bias += (nout * 8 + 1) * SW_BIAS_TILE_FLOATS;
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            x3_split<true>(acc[n], bhi[2 * n], blo[2 * n], bhi[2 * n + 1], blo[2 * n + 1], 0.f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (deform) return;
    f32x16 hv[4];
    x3_seg<4, 0, 16, SEG_BIAS, TERMS>(hv, bhi, blo, xs);
    {
        f32x16 demb;
        pe_dir(v0, v1, v2, h, demb);
        u32x4 dhi[2], dlo[2];
        x3_split<false>(demb, dhi[0], dlo[0], dhi[1], dlo[1]);
        x3_seg<4, 0, 2, SEG_ACC, TERMS>(hv, dhi, dlo, xs);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = relu1(hv[n][r]);
            s = x3_head_part(t, xs.bias + (o * 4 + n) * SW_BIAS_TILE_FLOATS, s);
        }
        rgb[o] = s + __shfl_xor(s, 32, 64) + hb[1 + o];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same canonical net with the split phase HIDDEN under the matrix pipe (software pipeline across layers).
// With k-block-major order layer L+1 needs tile t of layer L's output (k-blocks 2t, 2t+1) only when it reaches them, so
// the (hi, lo) split of tile t+1 - 8 pairs of values, ~10 VALU operations each - is spread over the 2*NT groups
// (6*NT MFMAs) that consume tile t.  Layer L's accumulators therefore stay fp32 while layer L+1 runs: two accumulator
// sets take turns (256 registers), but only a two-tile window of B operands exists at any time (32 registers).
// The accumulators of a layer start from its bias tiles by way of the first MFMA's srcC (read from LDS one group ahead),
// not by a 128-register init phase.
struct X3Win { u32x4 hi[2], lo[2]; };     // the two k-blocks of ONE source tile

// one pair (values 2q, 2q+1) of source tile `v` -> window w; optionally the fp32 head: s += wt[..] * relu(value)
template <int Q>
__device__ __forceinline__ void x3_split_pair(const f32x16& v, float floor, X3Win& w, const float* head_w, float& head_s) {
    const float a = x3_floor(v[2 * Q], floor), b = x3_floor(v[2 * Q + 1], floor);
    const unsigned h2 = x3_cvt_pk(a, b);
    const float ra = a - __uint_as_float(h2 << 16), rb = b - __uint_as_float(h2 & 0xffff0000u);
    const unsigned l2 = x3_cvt_pk(ra, rb);
    w.hi[Q >> 2][Q & 3] = h2; w.lo[Q >> 2][Q & 3] = l2;
    if (head_w) { head_s = fmaf(head_w[2 * Q], a, head_s); head_s = fmaf(head_w[2 * Q + 1], b, head_s); }   // wave-uniform branch
}
template <int Q0, int Q1>
__device__ __forceinline__ void x3_split_pairs(const f32x16& v, float floor, X3Win& w, const float* head_w, float& head_s) {
    if constexpr (Q0 < Q1) {
        x3_split_pair<Q0>(v, floor, w, head_w, head_s);
        x3_split_pairs<Q0 + 1, Q1>(v, floor, w, head_w, head_s);
    }
}

// groups of one 256-wide layer: dst[n] = bias_n + sum_kb W(n,kb) . split(src)[kb];  NT output tiles (8, or 4 for the view layer)
template <int G, int NT, int TERMS>
__device__ __forceinline__ void x3_pgroups(f32x16 (&dst)[NT], const f32x16 (&src)[8], float floor, X3Win& cur, X3Win& nxt, f32x16& binit,
                                           const float* head_w, float& head_s, XStream& xs) {
    constexpr int NG = NT * 16;
    if constexpr (G < NG) {
        constexpr int g = G % X3_CHUNK_GROUPS, kb = G / NT, n = G % NT, t = kb / 2, c = kb % 2, j = G % (2 * NT);
        if constexpr (g == X3_CHUNK_GROUPS - 1 - X3_AHEAD) x3_advance(xs);
        u32x4 nhi, nlo;
        if constexpr (g + X3_AHEAD >= X3_CHUNK_GROUPS) x3_read(xs.rd_next + (g + X3_AHEAD - X3_CHUNK_GROUPS) * X3_GROUP_BYTES, nhi, nlo);
        else x3_read(xs.rd + (g + X3_AHEAD) * X3_GROUP_BYTES, nhi, nlo);
        f32x16 bnext;
        if constexpr (kb == 0 && n + 1 < NT) {                  // the next tile's bias, one group ahead of the MFMA that starts from it
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xs.bias + (n + 1) * SW_BIAS_TILE_FLOATS + 4 * q);
                bnext[4 * q + 0] = v[0]; bnext[4 * q + 1] = v[1]; bnext[4 * q + 2] = v[2]; bnext[4 * q + 3] = v[3];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kb == 0) dst[n] = x3_mfma<TERMS>(xs.ahi, xs.alo, cur.hi[c], cur.lo[c], binit);
        else dst[n] = x3_mfma<TERMS>(xs.ahi, xs.alo, cur.hi[c], cur.lo[c], dst[n]);
        // the slice of tile t+1's split that rides behind this group's MFMAs (VALU issue slots the matrix pipe leaves free)
        if constexpr (t + 1 < 8) {
            constexpr int per = 8 / (2 * NT) > 0 ? 8 / (2 * NT) : 1;          // pairs per group: NT=4 -> 1; NT=8 -> one every 2nd group
            if constexpr (2 * NT <= 8) x3_split_pairs<j * per, j * per + per>(src[t + 1], floor, nxt, head_w ? head_w + (t + 1) * SW_BIAS_TILE_FLOATS : nullptr, head_s);
            else if constexpr (j % 2 == 1) x3_split_pairs<j / 2, j / 2 + 1>(src[t + 1], floor, nxt, head_w ? head_w + (t + 1) * SW_BIAS_TILE_FLOATS : nullptr, head_s);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            constexpr int part = (g - (X3_CHUNK_GROUPS - 1 - X3_AHEAD) + X3_CHUNK_GROUPS) % X3_CHUNK_GROUPS;
            if constexpr (part < 4) {
                x3_issue_part(xs, part);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (g == X3_CHUNK_GROUPS - 1) xs.rd = xs.rd_next;
#if X3_AHEAD == 2
        xs.ahi = xs.a2hi; xs.alo = xs.a2lo; xs.a2hi = nhi; xs.a2lo = nlo;
#else
        xs.ahi = nhi; xs.alo = nlo;
#endif
        if constexpr (kb == 0 && n + 1 < NT) binit = bnext;
        if constexpr (j == 2 * NT - 1 && t + 1 < 8) cur = nxt;
        x3_pgroups<G + 1, NT, TERMS>(dst, src, floor, cur, nxt, binit, head_w, head_s, xs);
    }
}

// one layer: dst = W . act(src) + b, act = max(., floor).  head_w != NULL: also head_s = sum_f head_w[f] * act(src)[f] (this
// lane's 128 features; bias-style weight tiles).  Consumes NT bias tiles at xs.bias + bias_skip tiles.
template <int NT, int TERMS>
__device__ __forceinline__ void x3_player(f32x16 (&dst)[NT], const f32x16 (&src)[8], float floor, const float* head_w, float& head_s,
                                          int bias_skip, XStream& xs) {
    X3Win cur, nxt;
    xs.bias += bias_skip * SW_BIAS_TILE_FLOATS;
    f32x16 binit;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xs.bias + 4 * q);
        binit[4 * q + 0] = v[0]; binit[4 * q + 1] = v[1]; binit[4 * q + 2] = v[2]; binit[4 * q + 3] = v[3];
    }
    x3_split_pairs<0, 8>(src[0], floor, cur, head_w, head_s);        // tile 0 up front: the only exposed part of the split
    x3_pgroups<0, NT, TERMS>(dst, src, floor, cur, nxt, binit, head_w, head_s, xs);
    xs.bias += NT * SW_BIAS_TILE_FLOATS;
}

template <int TERMS>
__device__ __forceinline__ void x3_canon_pipe(float px, float py, float pz, int h, const float* lds_dir, int lane,
                                              float& sigma, float (&rgb)[3], XStream& xs) {
    f32x16 A[8], B[8];
    float dummy = 0.f;
    {   // pts_linears[0] on gamma(x): the accumulators start from the bias tiles (the one init phase left)
        f32x16 emb[2];
        pe_pos(px, py, pz, h, emb);
        u32x4 ehi[4], elo[4];
        x3_split<false>(emb[0], ehi[0], elo[0], ehi[1], elo[1]);
        x3_split<false>(emb[1], ehi[2], elo[2], ehi[3], elo[3]);
        x3_seg<8, 0, 4, SEG_BIAS, TERMS>(A, ehi, elo, xs);
    }
    float hs = 0.f;
    const float* hb = nullptr;
    // layer pairs (1,2) (3,4) (5,6): odd layers A -> B, even layers B -> A; then layer 7 (A -> B) and the view layer on relu(h_7)
    // with feature_linear folded into its weights (round 4) - alpha_linear rides on that layer's split of h_7
#pragma nounroll
    for (int i = 0; i < 3; ++i) {
        x3_player<8, TERMS>(B, A, 0.f, nullptr, dummy, 0, xs);
        if (i == 2) {                                               // layer 5: ... then gamma(x) (model.py:45-46)
            f32x16 emb[2];
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
            pe_pos(px, py, pz, h, emb);
            u32x4 ehi[4], elo[4];
            x3_split<false>(emb[0], ehi[0], elo[0], ehi[1], elo[1]);
            x3_split<false>(emb[1], ehi[2], elo[2], ehi[3], elo[3]);
            x3_seg<8, 0, 4, SEG_ACC, TERMS>(B, ehi, elo, xs);
        }
        x3_player<8, TERMS>(A, B, 0.f, nullptr, dummy, 0, xs);
    }
    x3_player<8, TERMS>(B, A, 0.f, nullptr, dummy, 0, xs);             // layer 7
    const float* hw = xs.bias;                                      // alpha_linear: 8 weight tiles + the head-bias tile, in front of b_vf
    hb = xs.bias + 8 * SW_BIAS_TILE_FLOATS;
    f32x16 hv[4];
    x3_player<4, TERMS>(hv, B, 0.f, hw, hs, 9, xs);                    // views_linears[0] . feature_linear (folded) on relu(h_7) ...
    hs += __shfl_xor(hs, 32, 64);
    sigma = hs + hb[0];
    {
        f32x16 demb;
        tile_fetch(lds_dir, lane, demb);
        u32x4 dhi[2], dlo[2];
        x3_split<false>(demb, dhi[0], dlo[0], dhi[1], dlo[1]);
        x3_seg<4, 0, 2, SEG_ACC, TERMS>(hv, dhi, dlo, xs);                  // ... then gamma(d)
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = relu1(hv[n][r]);
            s = x3_head_part(t, xs.bias + (o * 4 + n) * SW_BIAS_TILE_FLOATS, s);
        }
        rgb[o] = s + __shfl_xor(s, 32, 64) + hb[1 + o];
    }
}

// DirectTemporalNeRF with the same software pipeline: `deform` selects the deformation net (layer 0 also takes gamma(t);
// layers 1..7; head = _time_out on relu(h_7), computed on the spot - nothing follows to ride on) or the canonical net
// (as x3_canon_pipe).  One body for both (see x3_net_dn).
template <int TERMS>
__device__ __forceinline__ void x3_net_dn_pipe(float px, float py, float pz, float ft, bool deform, int h, float v0, float v1, float v2,
                                               float (&head)[3], float (&rgb)[3], XStream& xs) {
    f32x16 A[8], B[8];
    float dummy = 0.f;
    {
        f32x16 emb[2];
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
        pe_pos(px, py, pz, h, emb);
        u32x4 ehi[6], elo[6];
        x3_split<false>(emb[0], ehi[0], elo[0], ehi[1], elo[1]);
        x3_split<false>(emb[1], ehi[2], elo[2], ehi[3], elo[3]);
        if (deform) {
            f32x16 te;
            pe_time(ft, h, te);
            x3_split<false>(te, ehi[4], elo[4], ehi[5], elo[5]);
            x3_seg<8, 0, 6, SEG_BIAS, TERMS>(A, ehi, elo, xs);
        } else {
            x3_seg<8, 0, 4, SEG_BIAS, TERMS>(A, ehi, elo, xs);
        }
    }
    // layer pairs (1,2) (3,4) (5,6) in one loop body, then layer 7; only the canonical net goes on (the view layer)
#pragma nounroll
    for (int i = 0; i < 3; ++i) {
        x3_player<8, TERMS>(B, A, 0.f, nullptr, dummy, 0, xs);
        if (i == 2) {
            f32x16 emb[2];
            asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
            pe_pos(px, py, pz, h, emb);
            u32x4 ehi[4], elo[4];
            x3_split<false>(emb[0], ehi[0], elo[0], ehi[1], elo[1]);
            x3_split<false>(emb[1], ehi[2], elo[2], ehi[3], elo[3]);
            x3_seg<8, 0, 4, SEG_ACC, TERMS>(B, ehi, elo, xs);
        }
        x3_player<8, TERMS>(A, B, 0.f, nullptr, dummy, 0, xs);
    }
    x3_player<8, TERMS>(B, A, 0.f, nullptr, dummy, 0, xs);             // layer 7
    if (deform) {                                                   // dx = _time_out(relu(h_7)): 3 x 8 weight tiles + the head-bias tile
#pragma nounroll
        for (int o = 0; o < 3; ++o) {
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] = relu1(B[n][r]);
                s = x3_head_part(t, xs.bias + (o * 8 + n) * SW_BIAS_TILE_FLOATS, s);
                __builtin_amdgcn_sched_barrier(0);
            }
            s += __shfl_xor(s, 32, 64);
            const float v = s + xs.bias[24 * SW_BIAS_TILE_FLOATS + o];
            if (o == 0) head[0] = v; else if (o == 1) head[1] = v; else head[2] = v;
        }
        xs.bias += 25 * SW_BIAS_TILE_FLOATS;
        return;
    }
    float hs = 0.f;
    const float* hw = xs.bias;                                      // alpha_linear rides on the split of h_7 (x3_canon_pipe)
    const float* hb = xs.bias + 8 * SW_BIAS_TILE_FLOATS;
    f32x16 hv[4];
    x3_player<4, TERMS>(hv, B, 0.f, hw, hs, 9, xs);                    // the view layer with feature_linear folded in, on relu(h_7)
    hs += __shfl_xor(hs, 32, 64);
    head[0] = hs + hb[0];
    {
        f32x16 demb;
        pe_dir(v0, v1, v2, h, demb);
        u32x4 dhi[2], dlo[2];
        x3_split<false>(demb, dhi[0], dlo[0], dhi[1], dlo[1]);
        x3_seg<4, 0, 2, SEG_ACC, TERMS>(hv, dhi, dlo, xs);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = relu1(hv[n][r]);
            s = x3_head_part(t, xs.bias + (o * 4 + n) * SW_BIAS_TILE_FLOATS, s);
        }
        rgb[o] = s + __shfl_xor(s, 32, 64) + hb[1 + o];
    }
}
