#!/usr/bin/env python3
"""bench.py - rays/sec of the fused NeRF render path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config C2 of BASELINE.json / SURVEY.md 8d): lego-like 800x800 camera, N_rand = 4096
rays per GPU per step, 64 coarse + 128 fine samples, separate coarse and fine 8x256 nets with
seeded synthetic weights, use_viewdirs, white background, perturb = 0 - i.e. exactly one call
of the reference's render(..., rays=batch_rays, **render_kwargs_test) per step.  Ray origins /
directions are resident in HBM before the timed region.  For N > 1 every rank renders its own
4096-ray batch (weak scaling) and the step ends with ONE RCCL all-gather of the rendered pixels
[rgb, disp, acc].  fp32 end to end (v_mfma_f32_32x32x2_f32).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel - the fine render pass
(192 samples/ray) - timed with events on the launch stream inside the timed region;
`cpu_baseline` is the CPU oracle (oracle/nerf_oracle.py, kind "port") on the host cores."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

FLOP_PER_ROW = 2 * 593408           # MACs of one (ray,sample) row through the 8x256 net
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
N_RAND, N_SAMPLES, N_IMPORTANCE, HW = 4096, 64, 128, 800


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup quota, and by
    16 (the GPU box gives one GPU's job a 16-core share although it shows every core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    # SWNERF_BENCH_REHEARSAL=1: run the N>1 control flow on a box with fewer GPUs than ranks (ranks share
    # cards, gloo instead of RCCL, pixels staged through the host for the gather).  Not a measurement.
    rehearsal = os.environ.get("SWNERF_BENCH_REHEARSAL") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from swnerf import synth, model, embedder, render, parallel

    embed_fn, input_ch = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, input_ch_views = embedder.get_embedder(4, 3, 0)
    nets, sds_np = [], []
    for seed, ab in (synth.NET_COARSE, synth.NET_FINE):
        sd = synth.nerf_state_dict(seed, alpha_bias=ab)
        m = model.vallina_NeRF(D=8, W=256, input_ch=input_ch, input_ch_views=input_ch_views, output_ch=5,
                               skips=[4], use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(m.to(dev).eval())
        sds_np.append(sd)
    network_query_fn = lambda inputs, viewdirs, network_fn: render.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    K, c2w = synth.lego_camera(HW, HW)
    o_np, d_np = synth.pick_rays(HW, HW, K, c2w, N_RAND, seed=2 + rank)
    rays_o, rays_d = torch.from_numpy(o_np).to(dev), torch.from_numpy(d_np).to(dev)
    kw = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets[0], network_query_fn=network_query_fn,
              N_samples=N_SAMPLES, N_importance=N_IMPORTANCE, network_fine=nets[1], white_bkgd=True,
              perturb=0., raw_noise_std=0.)

    fine_events = []

    def hook(phase, n_rays, n_samples):
        if n_samples == N_SAMPLES + N_IMPORTANCE and hook.on:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))     # the stream the kernel is launched on
            fine_events.append(ev)
    hook.on = False
    render.PASS_HOOK = hook

    def step():
        rgb, disp, acc, _ = render.render(HW, HW, K, chunk=1024 * 32, rays=(rays_o, rays_d), **kw)
        px = torch.cat([rgb, disp[:, None], acc[:, None]], -1)
        if world > 1 and rehearsal:
            return parallel.gather_pixels(px.cpu()).to(dev)
        return parallel.gather_pixels(px) if world > 1 else px

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        hook.on = True
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        fence()
        dt = time.perf_counter() - t0
        hook.on = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.shape == (world * N_RAND, 5) and bool(torch.isfinite(out[:, :3]).all())

    ms_kernel = [fine_events[i].elapsed_time(fine_events[i + 1]) for i in range(0, len(fine_events), 2)]
    fine_ms = float(np.mean(ms_kernel)) if ms_kernel else float("nan")
    fine_flop = N_RAND * (N_SAMPLES + N_IMPORTANCE) * FLOP_PER_ROW
    achieved = fine_flop / (fine_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("fine_pass_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "rays/sec (64+128 samples/ray)", "value": world * N_RAND * args.steps / dt, "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, gloo; not a measurement)" if rehearsal else ""),
        "config": {"workload": "C2: lego-like 800x800 camera, N_rand=4096 rays/GPU/step, 64 coarse + 128 fine samples, "
                               "coarse+fine 8x256 nets (use_viewdirs), white_bkgd, perturb=0; render() forward "
                               "(get rays resident -> ray batch -> coarse pass -> resample -> fine pass)"
                               + ("; + RCCL all-gather of [rgb,disp,acc]" if world > 1 else ""),
                   "rays_per_step_per_gpu": N_RAND, "n_samples": N_SAMPLES, "n_importance": N_IMPORTANCE,
                   "parallelism": f"ray-sharded dp{world}"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                     "kernel": "render_pass_kernel<false> fine pass (4096 rays x 192 samples)",
                     "ms_per_launch": fine_ms, "flop_per_launch": fine_flop},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import nerf_oracle as O
        torch.set_num_threads(host_cores())
        cores = torch.get_num_threads()
        sd_c, sd_f = (O.to_torch_sd(s) for s in sds_np)
        rb = O.make_ray_batch(torch.from_numpy(o_np), torch.from_numpy(d_np), 2., 6.)
        with torch.no_grad():
            O.render_rays(rb[:256], sd_c, sd_f, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)      # warm-up
            # bounded sample: the whole batch once (needed for the PSNR), then repeat up to ~15 s / 5 reps
            reps, t0 = 0, time.perf_counter()
            while reps < 1 or (time.perf_counter() - t0 < 15.0 and reps < 5):
                ref = O.render_rays(rb, sd_c, sd_f, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)
                reps += 1
            cdt = time.perf_counter() - t0
        mse = float(((out[:N_RAND, :3].cpu() - ref["rgb_map"]) ** 2).mean())
        result["cpu_baseline"] = {"value": reps * N_RAND / cdt, "unit": "rays/s", "cores": cores, "kind": "port",
                                  "sample": f"{reps} x the same 4096-ray batch (64+128, both nets), oracle/nerf_oracle.py, "
                                            f"torch CPU {torch.__version__}, no_grad"}
        result["psnr_vs_cpu_render_db"] = float(-10 * np.log10(max(mse, 1e-20)))
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
