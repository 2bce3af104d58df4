// render_pass.h - the fused NeRF render pass for gfx950 (MI355X): ONE wavefront owns ONE ray.
//
// For each 32-sample tile of its ray the wave computes depths -> points -> positional encoding (VALU) ->
// [deformation MLP ->] canonical MLP (MFMA, registers: mlp_core.h) -> alpha compositing (wave scan), and after the
// last tile optionally the hierarchical resampling (inverse-CDF + rank merge in the wave's LDS slice).  Nothing
// per-sample touches HBM unless the caller asks for it (raw / weights / dx / z_out).
// Shared by the inference translation unit (render_kernels.hip, ring depth 8) and the training one
// (train_kernels.hip, ring depth 16, TRAIN = true: the same pass that also saves what the backward needs).
//
// Reference: render_rays nerf/run.py:316-422, d_nerf/run_dnerf.py:354-480; raw2outputs ray.py:155-198;
// sample_pdf ray.py:96-153; run_network nerf/run.py:73-87.
#pragma once
#include <cstdlib>
#include "mlp_kernels.h"
#include "mlp_core_x3.h"
#include "resample.h"

#define SW_LDS_SC 256                    // max coarse samples when resampling
#define SW_LDS_SORT 1024                 // max S + n_importance (padded to a power of two)
#define SW_LDS_WAVE_FLOATS (3 * SW_LDS_SC + SW_LDS_SORT)

struct PassDev {
    swnerf_pass_args a;
    const float* w0;        // weight stream this pass runs per tile
    const float* b0;        // its bias stream
    int nbias;              // floats in the bias stream (multiple of 32)
    int two_pass;           // 1: deformation net then canonical net
    int sort_n, sort_s;     // powers of two >= n_importance / >= n_samples (fallback sort of an unsorted list)
    // TRAIN only (swnerf_render_pass_train): what the backward needs.  Rows are PADDED to whole 32-sample tiles per ray:
    // row = (ray * ceil(S/32) + tile) * 32 + j; rows past S carry copies of the last sample and get zero gradients.
    float* act;             // [rows, SW_ACT_LD]   post-ReLU activations (column map: swnerf_common.h)
    float* bits;            // [rows/32, SW_MASK_TILE_FLOATS]   ReLU bit masks
    float* xs;              // [rows, SW_XS_LD]    the encodings gamma(x), gamma(d) in B-operand SLOT order (sw_xs_col)
    // TRAIN + DNERF: the same three for the deformation net (act_d uses the first 2048 columns; xs_d = gamma(x), gamma(t))
    float* act_d; float* bits_d; float* xs_d;
    // start-up shaping of a launch (fp32 inference pass; pass_startup() below): L2 warm-up of the weight stream and a skew
    // of the waves' start.  warm_steps = 0 / skew_mode = 0 switch them off.
    int dir_steps;          // SW_STEPS_DIR when the stream starts with the per-ray DIR prefix (nets with view directions, fp32), else 0
    int time_steps;         // SW_STEPS_TIME when the deformation net's TIME segment follows it (D-NeRF with the deformation pass), else 0
    int tb_off;             // floats: where the four waves' per-ray TIME tiles sit in the dynamic LDS (behind everything else)
    int warm_steps;         // 1-KiB steps of the weight stream to pull into the XCD's L2 at kernel start
    int warm_blocks;        // workgroups per XCD that share the warm-up (blocks b with b < 8 * warm_blocks take part)
    int skew_mode;          // 0 none | 1 per workgroup | 2 per wave
    int skew_unit;          // s_sleep argument of one skew class (units of 64 clocks)
};

__device__ __forceinline__ float wave32_sum(float v) {   // sum over the 32 lanes of each half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
    return v;
}

__device__ __forceinline__ float z_linear(const swnerf_pass_args& a, float near, float far, int s) {
    const float t = sw_linspace(0.f, 1.f, a.n_samples, s);
    if (!a.lindisp) return near * (1.f - t) + far * t;                       // nerf/run.py:363
    return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);                   // nerf/run.py:365
}

// depth of sample s (0 <= s < S) of this ray
__device__ __forceinline__ float z_sample(const swnerf_pass_args& a, int64_t ray, float near, float far, int s) {
    const int S = a.n_samples;
    if (a.z_vals) return a.z_vals[ray * S + s];
    const float zs = z_linear(a, near, far, s);
    if (!a.t_rand) return zs;
    // stratified jitter, nerf/run.py:369-383
    const float upper = (s < S - 1) ? .5f * (z_linear(a, near, far, s + 1) + zs) : zs;
    const float lower = (s > 0) ? .5f * (zs + z_linear(a, near, far, s - 1)) : zs;
    return lower + (upper - lower) * a.t_rand[ray * S + s];
}

// ---- start-up shaping ----------------------------------------------------------------------------------------
// Every launch starts with the weight stream cold in the 8 XCD L2s (they are written back / invalidated at kernel
// boundaries) and with all resident waves at stream position 0.  The wave that leads takes every L2 miss, the others
// catch up behind it, and from then on all 128 waves of an XCD ask for the SAME 1-KiB step at the same moment: half of
// the L2 channels serve all of them while the other half idle.  A launch of many rounds pays that once (about 24 us at
// 4096 rays), a single-round launch (1024 rays = one wave per SIMD) for its whole duration.  So, before anything else:
//   warm-up: the first-round waves of an XCD split the stream between them and pull it into L2 with LDS-DMA into a junk
//            slot (no registers, nothing to wait for except the in-order vmcnt of the ring prime that follows);
//   skew:    waves (or workgroups) start 0..15 ring steps apart, so that concurrent requests spread over the channels.
#define SW_WARM_MAX_PER_WAVE 48
__device__ __forceinline__ void pass_startup(const PassDev& P, const float* w0, float* junk, int lane, int wv) {
    const unsigned b = blockIdx.x;
    if (P.warm_steps > 0 && (int)(b >> 3) < P.warm_blocks) {
        const unsigned junk_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)junk);
        const int stride = P.warm_blocks * 4;
        int s = (int)(b >> 3) * 4 + wv;
#pragma nounroll
        for (int i = 0; i < SW_WARM_MAX_PER_WAVE && s < P.warm_steps; ++i, s += stride)
            ws_dma(reinterpret_cast<const char*>(w0) + (size_t)s * 1024, (unsigned)lane * 16u, junk_addr);
    }
    if (P.skew_mode) {
        const int k = (P.skew_mode == 1 ? (int)(b >> 3) : (int)(b >> 3) * 4 + wv) & 15;
#pragma nounroll
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(4);
        (void)P.skew_unit;
    }
}

// host side: fill the start-up fields for a launch of `grid_x` workgroups whose pass streams `steps` weight steps per tile.
// SWNERF_WARM / SWNERF_SKEW (experiments: tools/probe_small_batch.py) override the defaults.
static inline void pass_startup_args(PassDev& P, unsigned grid_x, int steps) {
    // read per launch (two getenv calls, ~100 ns): tests flip them between launches to show that the shaping changes no bit
    const char* ew = getenv("SWNERF_WARM");
    const char* es = getenv("SWNERF_SKEW");
    const int env_warm = ew ? atoi(ew) : 1, env_skew = es ? atoi(es) : 2;
    const unsigned first_round = grid_x < 256u ? grid_x : 256u;      // one workgroup per CU is resident
    P.warm_blocks = (int)(first_round / 8u);
    P.warm_steps = (env_warm && P.warm_blocks > 0) ? steps + SW_TAIL : 0;
    P.skew_mode = env_skew;
    P.skew_unit = 4;
}

// ------------------------------------------------------------------------------------------
// TRAIN, static net: the LDS bias region holds the canonical tiles alone (the deformation tiles' 11 KB are what lets the
// 16-deep ring of the training translation unit AND the resampling scratch fit into 160 KB).  TRAIN + DNERF keeps both
// tile sets and has no resampling scratch (the D-NeRF training pass runs on given depths: the coarse pass of the
// one-model configuration is a no_grad inference pass, d_nerf/run_dnerf.py:417-421).
template <bool DNERF, bool TRAIN> struct PassLds {
    // (TRAIN static: room for the larger of the canonical tile set, 97, and the no-view-direction set, 64 + 8 x 5 + 1 = 105)
    static constexpr int TRAIN_TILES = SW_NOVIEW_BIAS_TILES(SW_NOVIEW_MAX_OUT) > SW_CANON_BIAS_TILES ? SW_NOVIEW_BIAS_TILES(SW_NOVIEW_MAX_OUT) : SW_CANON_BIAS_TILES;
    static constexpr int BIAS = (TRAIN && !DNERF) ? TRAIN_TILES * SW_BIAS_TILE_FLOATS : SW_LDS_BIAS_FLOATS;
    static constexpr int FIXED = BIAS + 4 * SW_LDS_RING_FLOATS;
};
// PREC != 0 (bf16x3 / bf16 pass, mlp_core_x3.h): bias tiles | the workgroup's shared weight ring | per wave: gamma(d)
// tile + depth slots | per wave: resampling scratch
// (D-NeRF: both bias-tile sets, and no gamma(d) tile - x3_net_dn evaluates it - so that the resampling scratch still fits)
template <bool DNERF> struct X3Lds {
    static constexpr int BIAS = (DNERF ? SW_DEFORM_BIAS_TILES + SW_X3_CANON_BIAS_TILES : SW_X3_CANON_BIAS_TILES) * SW_BIAS_TILE_FLOATS;
    static constexpr int DIR = DNERF ? 0 : 16 * 64;
    static constexpr int WAVE = DIR + SW_ZSLOT_FLOATS;
    static constexpr int FIXED = BIAS + X3_RING_FLOATS + 4 * WAVE;
};

// PREC: 0 = fp32 MFMA (mlp_core.h, the parity path); 3 = bf16x3, 1 = plain bf16 (mlp_core_x3.h; static net, inference)
// VIEWS = false: the net without view directions (SWNERF_NET_NOVIEW; model.py:59-60, 8-column ray batch nerf/run.py:152-157):
// the trunk, then output_linear as VALU heads - no feature / view branch, no gamma(d).  Static net, fp32, inference.
template <bool DNERF, bool TRAIN = false, int PREC = 0, bool VIEWS = true>
__global__ void __launch_bounds__(256, 1) render_pass_kernel(PassDev P) {
    static_assert(PREC == 0 || !TRAIN, "the bf16 paths cover the inference passes");
    static_assert(VIEWS || (!DNERF && PREC == 0), "the no-view-direction variant is a static fp32 pass (inference or TRAIN)");
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    SW_STAMP(probe_start);
    const swnerf_pass_args& a = P.a;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ray_id = (int64_t)blockIdx.x * 4 + wv;
    if constexpr (PREC == 0)                     // junk slot = this wave's parked-encoding region: nothing is parked before ws_start's wait
        pass_startup(P, P.w0, lds_all + PassLds<DNERF, TRAIN>::BIAS + wv * SW_LDS_RING_FLOATS + SW_RING * SW_STEP_FLOATS, lane, wv);
    bias_to_lds(lds_all, P.b0, P.nbias);         // fp32 path: the only block barrier; waves are independent after it
    // PREC: the four waves share the weight ring and run in step, so a wave past the last ray follows along on the
    // last ray and stores nothing
    const bool ghost = PREC != 0 && ray_id >= a.n_rays;
    if (PREC == 0 && ray_id >= a.n_rays) return; // wave-uniform
    const int64_t ray = ghost ? a.n_rays - 1 : ray_id;
    const float* lds_bias = lds_all;
    float* lds_ring = PREC ? lds_all + X3Lds<DNERF>::BIAS
                           : lds_all + PassLds<DNERF, TRAIN>::BIAS + wv * SW_LDS_RING_FLOATS;
    float* lds_emb = lds_ring + SW_RING * SW_STEP_FLOATS;
    float* lds_x3w = lds_ring + X3_RING_FLOATS + wv * X3Lds<DNERF>::WAVE;
    float* lds_dir = PREC ? lds_x3w : lds_emb + 2 * 16 * 64;
    float* lds_vb = lds_emb + SW_EMB_LDS_FLOATS + SW_ZSLOT_FLOATS;          // fp32, VIEWS: the per-ray init tiles of the view layer
    const float* lds_tb = (DNERF && PREC == 0 && P.time_steps) ? lds_all + P.tb_off + wv * SW_TB_LDS_FLOATS : nullptr;   // ... of _time.0
    float* lds = PREC ? lds_all + X3Lds<DNERF>::FIXED + wv * SW_LDS_WAVE_FLOATS
                      : lds_all + PassLds<DNERF, TRAIN>::FIXED + wv * SW_LDS_WAVE_FLOATS;
    float* zc = lds;                             // [S]   depths of this pass
    float* wc = lds + SW_LDS_SC;                 // [S]   compositing weights
    float* cdf = lds + 2 * SW_LDS_SC;            // [S-1]
    float* srt = lds + 3 * SW_LDS_SC;            // [sort_n]

    const int S = a.n_samples;
    const bool resample = a.n_importance > 0;
    const float* rb = a.ray_batch + ray * a.cols;
    const float ox = rb[0], oy = rb[1], oz = rb[2], dx = rb[3], dy = rb[4], dz = rb[5];
    const float near = rb[6], far = rb[7];
    const float ft = (a.cols == 12) ? rb[8] : 0.f;
    const float v0 = VIEWS ? rb[a.cols - 3] : 0.f, v1 = VIEWS ? rb[a.cols - 2] : 0.f, v2 = VIEWS ? rb[a.cols - 1] : 0.f;
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);                  // ray.py:173

    WStream ws;
    XStream xs;
    if constexpr (PREC != 0) {
        if (!DNERF) {         // once per ray: the view-direction encoding, parked in LDS (see tile_park)
            f32x16 demb;
            pe_dir(v0, v1, v2, h, demb);
            tile_park(lds_dir, lane, demb);
        }
        x3_start(xs, reinterpret_cast<const char*>(P.w0), lds_bias, lds_ring, lane, wv);
    } else {
        // ring prime first (its wait also retires the warm-up DMAs into the junk slot, vmcnt being in order), THEN the
        // view-direction encoding is parked where the junk went; its ~300 VALU cycles run while steps 1..7 are in flight
        ws_start(ws, P.w0, lds_bias, lds_ring, lane);
        if constexpr (VIEWS) {
            // the stream's DIR prefix, once per ray: c = Wv[:, 256:] gamma(d) + b_vf -> the view layer's init tiles (mlp_core.h)
            f32x16 demb;
            pe_dir(v0, v1, v2, h, demb);
            if constexpr (TRAIN) tile_park(lds_dir, lane, demb);             // the training passes also store gamma(d) per row (xs)
            view_bias_tile(demb, lds_vb, lane, ws);
        }
        // D-NeRF with the deformation pass: the stream's TIME segment, once per ray: _time.0.bias + its gamma(t) columns (mlp_core.h)
        if constexpr (DNERF) {
            if (P.time_steps) time_bias_tile(ft, h, lds_all + P.tb_off + wv * SW_TB_LDS_FLOATS, lane, ws);
        }
    }

    const float* zrow = a.z_vals ? a.z_vals + ray * S : nullptr;
    const float* zslot = PREC ? lds_x3w + X3Lds<DNERF>::DIR : lds_emb + SW_EMB_LDS_FLOATS;
    const unsigned zslot_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)zslot);
    float pr = 0.f, pg = 0.f, pb = 0.f, pd = 0.f, pa = 0.f;
    double Tc = 1.0;                              // transmittance carried across tiles
    const int ntiles = (S + 31) >> 5;
#ifdef SW_PROBE
    unsigned long long probe_acc[4] = {0ull, 0ull, 0ull, 0ull};
    SW_STAMP(probe_loop0);
#endif
#pragma nounroll
    for (int tile = 0; tile < ntiles; ++tile) {
        SW_STAMP(pt0);
        const int s = tile * 32 + j;
        const bool live = s < S && !ghost;
        const int sc = live ? s : S - 1;
        float z, zn;
        if (zrow && tile > 0) {
            // depths given (fine pass): fetched by LDS-DMA while the previous tile's MLP ran - issued before that
            // tile's weight stream, so long landed - instead of a global load whose latency every tile would expose
            const float* zs = zslot + (tile & 1) * 64;
            z = zs[j];
            const float z1 = zs[j + 1];
            zn = (s + 1 < S) ? z1 : z;
        } else {
            z = z_sample(a, ray, near, far, sc);
            zn = (s + 1 < S) ? z_sample(a, ray, near, far, s + 1) : z;
        }
        if (zrow && tile + 1 < ntiles)
            lds_dma_dword(reinterpret_cast<const char*>(zrow), (unsigned)min(32 * (tile + 1) + lane, S - 1) * 4u,
                          zslot_addr + (unsigned)((tile + 1) & 1) * 256u);
        // pts = rays_o + rays_d * z  (two roundings, nerf/run.py:385)
        float px = ox + dx * z, py = oy + dy * z, pz = oz + dz * z;

        f32x16 emb[2], in[8], out[8];
        float head[3], rgb[3];
        float extra = 0.f;                          // VIEWS = false, out_ch == 5: the fifth channel of output_linear (only `raw` shows it)
        pe_pos(px, py, pz, h, emb);
        SW_STAMP(pt1);
        if (DNERF && TRAIN) {
            // deformation net, then the canonical net on gamma(x + dx) (model.py:128-151); both save what their dX
            // chains and weight-gradient GEMMs need, as side stores (see the static branch below).  Always both passes:
            // the t == 0 / zero_canonical case trains the canonical net alone through the static kernel.
            const int64_t tix = ray * ntiles + tile, prow = tix * 32 + j;
            float* act_row = P.act + prow * SW_ACT_LD + 4 * h;
            float* mask_tile = P.bits + tix * SW_MASK_TILE_FLOATS + lane * 4;
            float* xs_row = P.xs + prow * SW_XS_LD + 4 * h;
            f32x4 mb = {0.f, 0.f, 0.f, 0.f};
            f32x16 demb;
            // ONE call site for both nets (a runtime loop like the inference kernel's): two would unroll the 8-layer body
            // twice - 4800 static MFMAs, beyond what the instruction cache holds comfortably
#pragma nounroll
            for (int pass = 0; pass < 2; ++pass) {
                const bool dp = pass == 0;
                trunk_pass<true, true, true, false, true>(emb, lds_emb, ft, dp, h, in, out, head, ws,
                                             dp ? P.act_d + prow * SW_ACT_LD + 4 * h : act_row,
                                             dp ? P.bits_d + tix * SW_MASK_TILE_FLOATS + lane * 4 : mask_tile, dp, &mb,
                                             dp ? P.xs_d + prow * SW_XS_LD + 4 * h : xs_row, lds_tb);
                if (dp) {
                    const float ex = head[0], ey = head[1], ez = head[2];
                    if (live && h == 0) {
                        float* o = a.dx + (ray * S + s) * 3;
                        o[0] = ex; o[1] = ey; o[2] = ez;
                    }
                    px = px + ex; py = py + ey; pz = pz + ez;
                    pe_pos(px, py, pz, h, emb);
                    tile_fetch(lds_dir, lane, demb);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {demb[4 * g], demb[4 * g + 1], demb[4 * g + 2], demb[4 * g + 3]};
                        *reinterpret_cast<f32x4*>(xs_row + 64 + 8 * g) = v;
                    }
                }
            }
            canon_tail_train(in, lds_vb, rgb, ws.bias - SW_BIAS_TILE_FLOATS, ws, act_row, mask_tile, mb);
        } else if (DNERF && PREC != 0) {
            if constexpr (DNERF && PREC != 0) {
#pragma nounroll
                for (int pass = P.two_pass ? 0 : 1; pass < 2; ++pass) {
#ifdef X3_NO_PIPE
                    x3_net_dn<PREC>(px, py, pz, ft, pass == 0, h, v0, v1, v2, head, rgb, xs);
#else
                    x3_net_dn_pipe<PREC>(px, py, pz, ft, pass == 0, h, v0, v1, v2, head, rgb, xs);
#endif
                    if (pass == 0) {
                        const float ex = head[0], ey = head[1], ez = head[2];
                        if (a.dx && live && h == 0) {
                            float* o = a.dx + (ray * S + s) * 3;
                            o[0] = ex; o[1] = ey; o[2] = ez;
                        }
                        px = px + ex; py = py + ey; pz = pz + ez;        // re-embedded inside the canonical pass
                    }
                }
                if (!P.two_pass && a.dx && live && h == 0) {
                    float* o = a.dx + (ray * S + s) * 3;
                    o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
                }
                x3_rewind(xs, P.two_pass ? SW_X3_DEFORM_CHUNKS + SW_X3_CANON_CHUNKS : SW_X3_CANON_CHUNKS, lds_bias, lane);
            }
        } else if (DNERF) {
#pragma nounroll
            for (int pass = P.two_pass ? 0 : 1; pass < 2; ++pass) {
                trunk_pass<true, false, false, false, true>(emb, lds_emb, ft, pass == 0, h, in, out, head, ws, nullptr, nullptr, false, nullptr, nullptr, lds_tb);
                if (pass == 0) {
                    // dx = _time_out(h) (model.py:136,146-149)
                    const float ex = head[0], ey = head[1], ez = head[2];
                    if (a.dx && live && h == 0) {
                        float* o = a.dx + (ray * S + s) * 3;
                        o[0] = ex; o[1] = ey; o[2] = ez;
                    }
                    px = px + ex; py = py + ey; pz = pz + ez;
                    pe_pos(px, py, pz, h, emb);                              // re-embed (model.py:148-149)
                }
            }
            if (!P.two_pass && a.dx && live && h == 0) {
                float* o = a.dx + (ray * S + s) * 3;
                o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;                          // model.py:144-145
            }
        } else if (TRAIN && !VIEWS) {
            // the net without view directions under autograd: the trunk's side stores as below (gamma(x) with layer 0, h_l and
            // its mask with layer l+1), h7 stored on the spot (no segment follows it), then output_linear as VALU heads
            const int64_t prow = (ray * ntiles + tile) * 32 + j;
            float* act_row = P.act + prow * SW_ACT_LD + 4 * h;
            float* mask_tile = P.bits + (ray * ntiles + tile) * SW_MASK_TILE_FLOATS + lane * 4;
            float* xs_row = P.xs + prow * SW_XS_LD + 4 * h;
            f32x4 mb = {0.f, 0.f, 0.f, 0.f};
            trunk_pass<false, true, true, true>(emb, lds_emb, 0.f, false, h, in, out, head, ws, act_row, mask_tile, true, &mb, xs_row);
            float o5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            head_valu_rt<8>(in, ws, a.out_ch, o5);
            rgb[0] = o5[0] + ws.bias[0]; rgb[1] = o5[1] + ws.bias[1]; rgb[2] = o5[2] + ws.bias[2];
            head[0] = o5[3] + ws.bias[3]; head[1] = 0.f; head[2] = 0.f;
            extra = o5[4] + ws.bias[4];
        } else if (TRAIN) {
            // the same tile, and everything the backward needs goes out as side stores of the segments (mlp_core.h
            // SideStore): gamma(x) with layer 0, h_l and its ReLU mask with layer l+1 (h7 with the view layer)
            const int64_t prow = (ray * ntiles + tile) * 32 + j;
            float* act_row = P.act + prow * SW_ACT_LD + 4 * h;
            float* mask_tile = P.bits + (ray * ntiles + tile) * SW_MASK_TILE_FLOATS + lane * 4;
            float* xs_row = P.xs + prow * SW_XS_LD + 4 * h;
            f32x4 mb = {0.f, 0.f, 0.f, 0.f};
            f32x16 demb;
            tile_fetch(lds_dir, lane, demb);
#pragma unroll
            for (int g = 0; g < 4; ++g) {                                    // gamma(d): k-tile 2 of the slot-ordered row
                const f32x4 v = {demb[4 * g], demb[4 * g + 1], demb[4 * g + 2], demb[4 * g + 3]};
                *reinterpret_cast<f32x4*>(xs_row + 64 + 8 * g) = v;
            }
            trunk_pass<false, true, true>(emb, lds_emb, 0.f, false, h, in, out, head, ws, act_row, mask_tile, false, &mb, xs_row);
            canon_tail_train(in, lds_vb, rgb, ws.bias - SW_BIAS_TILE_FLOATS, ws, act_row, mask_tile, mb);
        } else if constexpr (PREC != 0) {
            head[1] = 0.f; head[2] = 0.f;
#ifdef X3_NO_PIPE                                   // the plain form (split phase between layers): experiments / reference
            x3_canon<PREC>(px, py, pz, h, lds_dir, lane, head[0], rgb, xs);
#else
            x3_canon_pipe<PREC>(px, py, pz, h, lds_dir, lane, head[0], rgb, xs);
#endif
            x3_rewind(xs, SW_X3_CANON_CHUNKS, lds_bias, lane);
        } else if constexpr (!VIEWS) {
            trunk_pass<false, false, false, true>(emb, lds_emb, 0.f, false, h, in, out, head, ws);
            // outputs = output_linear(h) (model.py:59-60): [rgb(3), sigma, (5th, unused by raw2outputs)]
            float o5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            const int nout = (a.out_ch == 5 && !a.raw) ? 4 : a.out_ch;              // the fifth channel only shows in `raw`
            head_valu_rt<8>(in, ws, nout, o5);
            ws.bias += (a.out_ch - nout) * 8 * SW_BIAS_TILE_FLOATS;                 // skip the weight tiles of a channel not evaluated
            rgb[0] = o5[0] + ws.bias[0]; rgb[1] = o5[1] + ws.bias[1]; rgb[2] = o5[2] + ws.bias[2];
            head[0] = o5[3] + ws.bias[3]; head[1] = 0.f; head[2] = 0.f;
            extra = o5[4] + ws.bias[4];
        } else {
            trunk_pass<false>(emb, lds_emb, 0.f, false, h, in, out, head, ws);
        }
        SW_STAMP(pt2);
        if (!TRAIN && PREC == 0 && VIEWS) canon_tail(in, lds_vb, rgb, ws.bias - SW_BIAS_TILE_FLOATS, ws);
        SW_STAMP(pt3);
        // back to the head of MAIN (behind the per-ray DIR prefix and its b_vf tiles)
        if constexpr (PREC == 0)                 // back to MAIN: behind the per-ray prefixes (DIR, and TIME with its 8 bias tiles)
            ws_rewind(ws, P.w0 + (P.dir_steps + P.time_steps) * SW_STEP_FLOATS,
                      lds_bias + ((P.dir_steps ? SW_DIR_BIAS_TILES : 0) + (P.time_steps ? 8 : 0)) * SW_BIAS_TILE_FLOATS, lane);

        // ---- raw2outputs on this tile (ray.py:155-198); both lane halves mirror each other
        const float c0 = rgb[0], c1 = rgb[1], c2 = rgb[2];
        float sg = head[0];
        if (a.raw && live && h == 0) {
            if (!VIEWS && a.out_ch == 5) {
                float* o = a.raw + (ray * S + s) * 5;
                o[0] = c0; o[1] = c1; o[2] = c2; o[3] = sg; o[4] = extra;
            } else {
                f32x4 r4 = {c0, c1, c2, sg};
                *reinterpret_cast<f32x4*>(a.raw + (ray * S + s) * 4) = r4;
            }
        }
        if (a.noise) sg += a.noise[ray * S + sc];
        float dist = (s + 1 < S) ? (zn - z) : 1e10f;
        dist = dist * dnorm;
        float alpha = 1.f - expf(-fmaxf(sg, 0.f) * dist);
        if (!live) alpha = 0.f;
        double ps = (double)(1.f - alpha + 1e-10f);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const double up = __shfl_up(ps, o, 32);
            if (j >= o) ps *= up;
        }
        double ex = __shfl_up(ps, 1, 32);
        if (j == 0) ex = 1.0;
        const float T = (float)(Tc * ex);                                    // exclusive cumprod (ray.py:188)
        Tc *= __shfl(ps, 31, 32);
        const float w = alpha * T;
        if (live) {
            if (a.weights && h == 0) a.weights[ray * S + s] = w;
            if (a.z_out && h == 0) a.z_out[ray * S + s] = z;
            if (resample && h == 0) { zc[s] = z; wc[s] = w; }
        }
        pr += w * (1.f / (1.f + expf(-c0)));
        pg += w * (1.f / (1.f + expf(-c1)));
        pb += w * (1.f / (1.f + expf(-c2)));
        pd += w * z;
        pa += w;
#ifdef SW_PROBE
        {
            SW_STAMP(pt4);
            probe_acc[0] += pt1 - pt0; probe_acc[1] += pt2 - pt1; probe_acc[2] += pt3 - pt2; probe_acc[3] += pt4 - pt3;
        }
#endif
    }

    pr = wave32_sum(pr); pg = wave32_sum(pg); pb = wave32_sum(pb);
    pd = wave32_sum(pd); pa = wave32_sum(pa);
    if (lane == 0 && !ghost) {
        if (a.rgb_map) {
            const float bg = a.white_bkgd ? (1.f - pa) : 0.f;                // ray.py:195-196
            a.rgb_map[ray * 3 + 0] = pr + bg;
            a.rgb_map[ray * 3 + 1] = pg + bg;
            a.rgb_map[ray * 3 + 2] = pb + bg;
        }
        if (a.depth_map) a.depth_map[ray] = pd;
        if (a.acc_map) a.acc_map[ray] = pa;
        if (a.disp_map) {
            const float q = pd / pa;                                         // NaN when acc == 0, kept (ray.py:192)
            a.disp_map[ray] = 1.f / ((q != q) ? q : fmaxf(1e-10f, q));
        }
    }
#ifdef SW_PROBE
    if (a.weights && lane == 0 && !ghost) {          // [sampling+encoding, trunk, tail, compositing, whole tile loop, prologue] cycles of this wave
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.weights + ray * S);
        const unsigned long long pend = sw_clock();
        o[0] = probe_acc[0]; o[1] = probe_acc[1]; o[2] = probe_acc[2]; o[3] = probe_acc[3]; o[4] = pend - probe_loop0; o[5] = probe_loop0 - probe_start;
        if constexpr (PREC != 0) { o[6] = xs.pc[0]; o[7] = xs.pc[1]; o[8] = xs.pc[2]; }
    }
#endif
    if (!resample || ghost) return;
#ifdef SW_PROBE
    const unsigned long long probe_rs0 = sw_clock();
#endif

    // ---- sample_pdf (ray.py:96-153) on bins = mid-points, weights[1:-1]; z_std; then sort (nerf/run.py:396-400, 416) as a rank
    // merge - the wave-level routines of resample.h, shared with the standalone op
    wave_lds_sync();
    const int Ni = a.n_importance;
    const double sm = wave_sample_pdf(wc + 1, S - 1, MidBins{zc}, a.u ? a.u + ray * Ni : nullptr, Ni, cdf, srt, lane);
    if (a.z_std) {
        const float sd = wave_zstd(sm, srt, Ni, lane);
        if (lane == 0) a.z_std[ray] = sd;
    }
    wave_rank_merge(zc, S, P.sort_s, srt, Ni, P.sort_n, a.z_fine + ray * (S + Ni), lane);
#ifdef SW_PROBE
    if (a.weights && lane == 0) {                    // [9] the resampling tail, [10] compositing epilogue .. its start
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.weights + ray * S);
        o[9] = sw_clock() - probe_rs0;
    }
#endif
}

