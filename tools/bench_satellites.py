#!/usr/bin/env python3
"""HBM-bound satellites of the render path (SURVEY.md 8a rows a1, a4, a6, a10, a11 and the compositing backward) at sizes
where they are bandwidth- rather than launch-bound: microseconds per launch (events on the launch stream, reps back to
back), algorithmic bytes, GB/s, fraction of the 8 TB/s spec and of the 6.29 TB/s a float4 copy reaches on this part
(MI355X_MICROARCH.md).  Through the C ABI with preallocated operands: nothing but the kernel is inside the timed region.

  python tools/bench_satellites.py [--json out.json] [--only name] [--reps N]

Under rocprofv3 (tools/profile_r04_satellites.sh) the same launches give kernel durations and FETCH_SIZE / WRITE_SIZE.
Reference: ray.py:10-38 (get_rays), nerf/run.py:137-158 (ray batch), embedder.py:33-42, ray.py:155-198 (raw2outputs),
ray.py:96-153 + nerf/run.py:396-400 (sample_pdf + sort)."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_SPEC, HBM_ACHIEVABLE = 8.0e12, 6.29e12


def run_satellites(dev, reps=20, only=None, quiet=False):
    """-> rows (dicts: name, kernel, bound, us, algorithmic_bytes, GBps, frac_of_8TBps, frac_of_achievable_6p29).  bench.py puts
    them into its record (extra.satellites)."""
    import types
    args = types.SimpleNamespace(reps=reps, only=only)
    import numpy as np
    import torch
    from swnerf import _lib, synth
    L = _lib.lib()
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = _lib.ptr
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    rnd = lambda *s: torch.rand(s, device=dev, generator=g)
    rows = []

    VALU = ("valu", "VALU-issue bound, not HBM bound: the vector ALUs are saturated (profiles/r04/satellites_pmc.md: VALUBusy >= 100 %, "
                    "counter bytes == algorithmic bytes) - exact expf / IEEE division / double-precision scans per sample")
    EMB = ("valu", "VALU-issue bound (VALUBusy 52-79 %, profiles/r04/satellites_pmc.md): a sin/cos pair per two output floats")

    def run(name, kernel, fn, nbytes, bound=("hbm", "")):
        note = bound[1]
        if args.only and args.only not in name:
            return
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        for _ in range(args.reps):
            fn()
        e1.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        bw = nbytes / (us * 1e-6)
        rows.append({"name": name, "kernel": kernel, "bound": bound[0], "us": us, "algorithmic_bytes": nbytes, "GBps": bw / 1e9,
                     "frac_of_8TBps": bw / HBM_SPEC, "frac_of_achievable_6p29": bw / HBM_ACHIEVABLE, "note": note})
        if not quiet:
            print(f"{name:64s} {us:9.1f} us  {nbytes / 1e6:9.1f} MB  {bw / 1e9:8.1f} GB/s  {bw / HBM_SPEC:6.1%} of spec  {bw / HBM_ACHIEVABLE:6.1%} of achievable",
                  flush=True)

    # ---- get_rays (a1): 800x800 as C4 renders it, and a 4000x4000 grid where launch ramp no longer shows
    K, c2w = synth.lego_camera(800, 800)
    m = np.ascontiguousarray(np.asarray(c2w)[:3, :4], dtype=np.float32)
    mp = m.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    for H in (800, 4000):
        n = H * H
        ro, rd = torch.empty((n, 3), device=dev), torch.empty((n, 3), device=dev)
        f = float(K[0][0]) * H / 800
        run(f"get_rays {H}x{H} (rays_o + rays_d written)", "get_rays_kernel",
            lambda: _lib.check(L.swnerf_get_rays(H, H, f, f, H * 0.5, H * 0.5, 0, mp, 0, n, p(ro), p(rd), st()), "get_rays"), n * 24)
    # ---- pack_ray_batch (a4) on the 800x800 frame and on 16 M rays
    for n in (640000, 16000000):
        ro, rd = rnd(n, 3) - 0.5, rnd(n, 3) - 0.5
        rb = torch.empty((n, 11), device=dev)
        run(f"pack_ray_batch {n} rays -> [N,11]", "pack_rays_kernel",
            lambda: _lib.check(L.swnerf_pack_ray_batch(p(ro), p(rd), n, 2., 6., 0, 0., 0, 800, 800, 1111., p(rb), st()), "pack"), n * (24 + 44))
        rb = torch.empty((n, 12), device=dev)
        run(f"pack_ray_batch {n} rays -> [N,12] (D-NeRF, NDC off)", "pack_rays_kernel",
            lambda: _lib.check(L.swnerf_pack_ray_batch(p(ro), p(rd), n, 2., 6., 1, .5, 0, 800, 800, 1111., p(rb), st()), "pack"), n * (24 + 48))
        del ro, rd, rb
    # ---- embed (a6): the C2 fine pass's rows
    for M, d, Lb in ((786432, 3, 10), (786432, 3, 4), (786432, 1, 10), (8388608, 3, 10), (8388608, 1, 10)):   # ... and sizes where launch ramp no longer shows
        x = rnd(M, d) * 12 - 6
        C = d * (1 + 2 * Lb)
        out = torch.empty((M, C), device=dev)
        run(f"embed {M} x {d} -> {C} (L={Lb})", "embed_kernel", lambda: _lib.check(L.swnerf_embed(p(x), M, d, Lb, p(out), st()), "embed"), M * 4 * (d + C), EMB)
        del x, out
    # ---- raw2outputs (a10) forward / backward on the 800x800 frame at S = 192 (and S = 64)
    for S in (192, 64):
        N = 640000
        raw = rnd(N, S, 4) * 4 - 2
        z = torch.sort(rnd(N, S) * 4 + 2, -1).values.contiguous()
        rd = rnd(N, 3) - 0.5
        o3, o1a, o1b, o1c, w = (torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev),
                                torch.empty(N, device=dev), torch.empty((N, S), device=dev))
        run(f"raw2outputs {N} x {S} (all five outputs)", "raw2outputs_kernel",
            lambda: _lib.check(L.swnerf_raw2outputs(p(raw), p(z), p(rd), None, N, S, 1, p(o3), p(o1a), p(o1b), p(w), p(o1c), st()), "r2o"),
            N * (S * 24 + 12 + 24), VALU)
        run(f"raw2outputs {N} x {S} (maps only, no weights)", "raw2outputs_kernel",
            lambda: _lib.check(L.swnerf_raw2outputs(p(raw), p(z), p(rd), None, N, S, 1, p(o3), p(o1a), p(o1b), None, p(o1c), st()), "r2o"),
            N * (S * 20 + 12 + 24), VALU)
        g3, g1 = rnd(N, 3), rnd(N)
        d_raw = torch.empty_like(raw)
        run(f"raw2outputs backward {N} x {S} (d rgb, d disp, d acc -> d raw)", "raw2outputs_bwd_kernel",
            lambda: _lib.check(L.swnerf_raw2outputs_backward(p(raw), p(z), p(rd), None, N, S, 1, p(g3), p(g1), p(g1), None, None, p(d_raw), st()), "r2ob"),
            N * (S * 20 + S * 16 + 12 + 20), VALU)
        del raw, z, w, d_raw
    # ---- sample_pdf + sort (a11): 63 bins -> 128 samples -> 192 sorted depths, the frame's rays
    N, S, Ni = 640000, 64, 128
    zc = torch.sort(rnd(N, S) * 4 + 2, -1).values.contiguous()
    bins = (.5 * (zc[:, 1:] + zc[:, :-1])).contiguous()
    wts = rnd(N, S - 2)
    smp, zs, zstd = torch.empty((N, Ni), device=dev), torch.empty((N, S + Ni), device=dev), torch.empty(N, device=dev)
    run(f"sample_pdf {N} x ({S - 1} bins -> {Ni}), det", "sample_pdf_kernel",
        lambda: _lib.check(L.swnerf_sample_pdf(p(bins), p(wts), N, S - 1, Ni, None, p(smp), None, 0, None, None, st()), "spdf"),
        N * 4 * ((S - 1) + (S - 2) + Ni), VALU)
    run(f"sample_pdf + sort {N} x ({S - 1} bins -> {Ni} -> {S + Ni}), det, z_std", "sample_pdf_kernel",
        lambda: _lib.check(L.swnerf_sample_pdf(p(bins), p(wts), N, S - 1, Ni, None, p(smp), p(zc), S, p(zs), p(zstd), st()), "spdf"),
        N * 4 * ((S - 1) + (S - 2) + S + Ni + (S + Ni) + 1), VALU)
    u = rnd(N, Ni)
    run(f"sample_pdf + sort {N} x ({S - 1} bins -> {Ni} -> {S + Ni}), random u", "sample_pdf_kernel",
        lambda: _lib.check(L.swnerf_sample_pdf(p(bins), p(wts), N, S - 1, Ni, p(u), p(smp), p(zc), S, p(zs), p(zstd), st()), "spdf"),
        N * 4 * ((S - 1) + (S - 2) + S + Ni + Ni + (S + Ni) + 1), VALU)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    rows = run_satellites(torch.device("cuda:0"), args.reps, args.only)
    if args.json:
        json.dump({"hbm_spec_Bps": HBM_SPEC, "hbm_achievable_Bps": HBM_ACHIEVABLE, "reps": args.reps, "rows": rows}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
