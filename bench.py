#!/usr/bin/env python3
"""bench.py - rays/sec of the fused NeRF render path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config C2|C4|C5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT a launcher (WORLD_SIZE unset) and --gpus N > 1, this process becomes a launcher: it compiles the
library if stale (hipcc only), starts N fresh rank processes (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set)
BEFORE anything in it touches HIP or torch.cuda, relays rank 0's JSON line and exits with the worst child code.

Workloads (BASELINE.json configs / SURVEY.md 8d):
  C2 (default, the headline; weak scaling): lego-like 800x800 camera, N_rand = 4096 rays per GPU per step, 64 coarse
     + 128 fine samples, separate coarse and fine 8x256 nets with seeded synthetic weights, use_viewdirs, white
     background, perturb = 0 - one call of the reference's render(..., rays=batch_rays, **render_kwargs_test) per
     step; ray origins / directions resident in HBM before the timed region; for N > 1 every rank renders its own
     batch and the step ends with ONE RCCL all-gather of the rendered pixels [rgb, disp, acc].
  C4 (strong scaling): the whole 800x800 frame (640 000 rays) per step, the reference's render-only entry
     (nerf/run.py:557-571): rank r generates the rays of its contiguous row-major shard (get_rays on the range),
     renders them (64+128, two nets) and the step ends with the all-gather of the frame.
  C5 (strong scaling): D-NeRF 400x400 frame (160 000 rays) at t = 0.5, one DirectTemporalNeRF (deformation +
     canonical net per sample, d_nerf/run_dnerf.py:553-566), same sharding.
  train (weak scaling; NOT the BASELINE metric - it exists so that the data-parallel training collective has a driver-shaped
     entry): the reference's training step (nerf/run.py:684-708) on the C2 shape - render(4096 rays, 64+128, two nets,
     perturb=1) -> mse(rgb) + mse(rgb0) -> backward (fused passes) -> Adam; for N > 1 (or --collective always) the gradients
     live in swnerf.parallel.GradBucket: one in-place RCCL all-reduce per net, issued async from autograd's hooks (the fine
     net's under the coarse pass's backward), waited for before optimizer.step().
fp32 end to end (v_mfma_f32_32x32x2_f32).

--collective always: the N = 1 run joins an RCCL process group of one rank as well (init_process_group("nccl"), the
all_gather_into_tensor of the pixels on device tensors at the end of every step, the all_reduce(MAX) of the timing) -
the code path of the N > 1 runs on a box with one GPU.  Default "auto": collectives only when N > 1.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel - the fine render pass (192 samples/ray) -
timed with events on the launch stream inside the timed region; `cpu_baseline` is the CPU oracle
(oracle/nerf_oracle.py, kind "port") on the host cores (N = 1 only); `extra.configs` (N = 1 only, measured AFTER and
outside the headline's timed region) carries the other configs and the training step with their own roofline
fractions, so that every number DESIGN.md quotes sits in a driver-run record."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

FLOP_PER_ROW = 2 * 593408           # MACs of one (ray,sample) row through the 8x256 net AS THE REFERENCE RUNS IT
FLOP_PER_ROW_DEFORM = 2 * 497152    # ... through the deformation net (D-NeRF, t != 0)
# Round 4: feature_linear (256 x 256, no activation: model.py:49-53) is folded into views_linears.0 at pack time, so the
# kernels EXECUTE 65 536 MACs per row fewer than the reference's algorithm.  Every roofline fraction in this file is
# executed FLOPs / time / peak (it cannot exceed 1); the reference-algorithmic rate rides beside it as algorithmic_tflops.
# ... and the view layer's gamma(d) columns (27 x 128, padded to one 32-slot k-tile = 4096 MACs per row) are a per-RAY constant
# that the fused passes evaluate once per ray and pass (64 MFMAs = 131 072 MACs) instead of once per row.
FLOP_EXEC_PER_ROW = 2 * (593408 - 65536 - 4096)
FLOP_EXEC_PER_RAY_PASS = 2 * 131072
# D-NeRF: _time.0's gamma(t) columns (21 x 256 = 5376 MACs per row) are a per-RAY constant too (one frame time per ray): the fused
# passes evaluate the stream's TIME segment once per ray and pass (128 MFMAs = 262 144 MACs) and start every tile from that tile
FLOP_EXEC_PER_ROW_DEFORM = 2 * (497152 - 5376)
FLOP_EXEC_PER_RAY_PASS_DEFORM = 2 * 262144
# training: forward (executed) + dX chain (RGB^T 16 + W_vf^T 128 + 7 x 256 weight steps x 4 MFMAs x 2048 MACs / 32 rows
# = 495 616 MACs per row: no layer-0 / view-direction input gradients) + dW (the forward's shapes with G = d pre_hv^T h7
# in place of feature_linear's GEMM = 527 872); the reference-algorithmic figure stays 3 x the forward
FLOP_EXEC_TRAIN_PER_ROW = 2 * (523776 + 495616 + 527872)
FLOP_EXEC_TRAIN_DEFORM_PER_ROW = 2 * (497152 - 5376 + 7 * 65536 + 497152) # deformation net: forward (executed) + L7..L1 transposed + dW
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md); only used by --precision bf16x3 lines
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
N_RAND, N_SAMPLES, N_IMPORTANCE = 4096, 64, 128
DEFAULT_STEPS = {"C2": (50, 5), "C4": (5, 1), "C5": (10, 2), "train": (20, 3)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["C2", "C4", "C5", "train"], default="C2",
                    help="C2 (default): the headline; train: the reference's TRAINING step on the C2 shape (not the BASELINE metric)")
    ap.add_argument("--collective", choices=["auto", "always"], default="auto",
                    help="always: also at N = 1 join an RCCL group (world 1) and end every step with the real all-gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra.configs block")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16x3-fine"], default="fp32",
                    help="arithmetic of the fused passes; fp32 (default) is the metric BASELINE.json names and the only one the "
                         "driver measures - the others are the opt-in bf16x3 paths (DESIGN.md 7c), labelled as such in the line")
    args = ap.parse_args()
    ds, dw = DEFAULT_STEPS[args.config]
    args.steps = ds if args.steps is None else args.steps
    args.warmup = dw if args.warmup is None else args.warmup
    return args


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


# ------------------------------------------------------------------------------------------- launcher
def visible_gpu_count(nodes="/sys/class/kfd/kfd/topology/nodes", dri="/dev/dri"):
    """GPUs this process tree can use, WITHOUT importing torch or touching HIP (the launcher parent must never
    initialise a GPU: a process that has may not start programs on this pool).  KFD topology nodes with SIMDs whose DRM
    render node is present and accessible (a container sees every node of the host in sysfs but only its own cards in
    /dev/dri), then the *_VISIBLE_DEVICES lists."""
    import glob
    n = 0
    have_dri = os.path.isdir(dri)
    for prop in sorted(glob.glob(os.path.join(nodes, "*", "properties"))):
        try:
            kv = dict(l.split(None, 1) for l in open(prop).read().splitlines() if " " in l)
        except OSError:
            continue
        if int(kv.get("simd_count", "0")) <= 0:
            continue                                     # a CPU node
        minor = int(kv.get("drm_render_minor", "-1"))
        if have_dri and minor >= 0 and not os.access(os.path.join(dri, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch(args):
    """--gpus N > 1 without a launcher: become one.  Nothing here may initialise the GPU (a process that has must
    not start programs on this pool): hipcc through subprocess, the GPU count from sysfs (no torch in this process),
    then N children, each a fresh interpreter.  The children are polled: the first one that fails takes its siblings
    down (they would otherwise sit in init_process_group / a collective until the 10-30 min backend timeout) and its
    exit code becomes the launcher's.  Never re-execs."""
    import __graft_entry__
    __graft_entry__.compile_library()
    rehearsal = os.environ.get("SWNERF_BENCH_REHEARSAL") == "1"
    have = visible_gpu_count()
    if have < args.gpus and not rehearsal:
        print(f"[bench] --gpus {args.gpus} but only {have} GPU(s) visible; refusing to report a {args.gpus}-GPU number "
              f"(SWNERF_BENCH_REHEARSAL=1 runs the control flow with ranks sharing cards over gloo)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0 owns stdout (the JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                print(f"[bench] rank process {procs.index(p)} exited with {code}: stopping the other ranks", file=sys.stderr)
                for q in live:
                    q.terminate()
                t_kill = time.time() + 15.0
                while any(q.poll() is None for q in live) and time.time() < t_kill:
                    time.sleep(0.2)
                for q in live:
                    if q.poll() is None:
                        q.kill()
    return rc


# --------------------------------------------------------------------------------------------- worker
def build_scene(cfg, dev, rank):
    """nets, render kwargs and the camera of one workload; everything seeded (swnerf/synth.py)."""
    import torch
    from swnerf import synth, model, embedder, render, render_dnerf
    embed_fn, input_ch = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, input_ch_views = embedder.get_embedder(4, 3, 0)
    sc = {"sds_np": []}
    if cfg == "train":
        cfg = "C2"
    if cfg == "C5":
        embedtime_fn, input_ch_time = embedder.get_embedder(10, 1, 0)
        sd = synth.dnerf_state_dict(synth.NET_DNERF[0], alpha_bias=synth.NET_DNERF[1])
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):     # the factory prints "NeRF type selected: ..." like the reference's (model.py:216): stdout is ONE JSON line
            net = model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=input_ch, output_ch=5, skips=[4],
                                         input_ch_views=input_ch_views, input_ch_time=input_ch_time, use_viewdirs=True,
                                         embed_fn=embed_fn, zero_canonical=True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        net = net.to(dev).eval()
        sc["sds_np"] = [sd]
        query = lambda inputs, viewdirs, ts, network_fn: render_dnerf.run_network(
            inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn,
            netchunk=1024 * 64, embd_time_discr=True)
        sc["H"] = sc["W"] = 400
        sc["kw"] = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=net, network_query_fn=query, N_samples=N_SAMPLES,
                        N_importance=N_IMPORTANCE, network_fine=None, white_bkgd=True, perturb=0., raw_noise_std=0.)
        sc["frame_time"] = 0.5
        sc["flop_per_ray"] = (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * (FLOP_PER_ROW + FLOP_PER_ROW_DEFORM)
        sc["flop_per_fine_row"] = FLOP_PER_ROW + FLOP_PER_ROW_DEFORM
        sc["exec_per_ray"] = ((N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * (FLOP_EXEC_PER_ROW + FLOP_EXEC_PER_ROW_DEFORM)
                              + 2 * (FLOP_EXEC_PER_RAY_PASS + FLOP_EXEC_PER_RAY_PASS_DEFORM))
        sc["exec_per_fine_row"] = FLOP_EXEC_PER_ROW + FLOP_EXEC_PER_ROW_DEFORM
        sc["exec_per_ray_pass"] = FLOP_EXEC_PER_RAY_PASS + FLOP_EXEC_PER_RAY_PASS_DEFORM
        sc["kernel"] = "render_pass_kernel<true>"
    else:
        nets = []
        for seed, ab in (synth.NET_COARSE, synth.NET_FINE):
            sd = synth.nerf_state_dict(seed, alpha_bias=ab)
            m = model.vallina_NeRF(D=8, W=256, input_ch=input_ch, input_ch_views=input_ch_views, output_ch=5,
                                   skips=[4], use_viewdirs=True)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            nets.append(m.to(dev).eval())
            sc["sds_np"].append(sd)
        query = lambda inputs, viewdirs, network_fn: render.run_network(
            inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
        sc["H"] = sc["W"] = 800
        sc["kw"] = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets[0], network_query_fn=query,
                        N_samples=N_SAMPLES, N_importance=N_IMPORTANCE, network_fine=nets[1], white_bkgd=True, perturb=0.,
                        raw_noise_std=0.)
        sc["frame_time"] = None
        sc["flop_per_ray"] = (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * FLOP_PER_ROW
        sc["flop_per_fine_row"] = FLOP_PER_ROW
        sc["exec_per_ray"] = (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * FLOP_EXEC_PER_ROW + 2 * FLOP_EXEC_PER_RAY_PASS
        sc["exec_per_fine_row"] = FLOP_EXEC_PER_ROW
        sc["exec_per_ray_pass"] = FLOP_EXEC_PER_RAY_PASS
        sc["kernel"] = "render_pass_kernel<false>"
    sc["K"], sc["c2w"] = synth.lego_camera(sc["H"], sc["W"])
    return sc


def extra_configs(dev):
    """The other BASELINE configs and the training step on this GPU, wall-clock per call through the Python mirrors
    (so Python, get_rays and ray-batch packing are inside), each with its own fraction of the fp32-MFMA roofline."""
    import numpy as np
    import torch
    from swnerf import synth, render, render_dnerf, parallel
    rows = []
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    from oracle import nerf_oracle as O

    def timeit(name, fn, n_rays, flop_per_ray, reps, grad=False, ref=None, exec_per_ray=None):
        """flop_per_ray: the reference-algorithmic FLOPs; exec_per_ray: the FLOPs the kernels execute (None: the same).
        ref: None, or callable(first 256 rows of fn()'s rgb on the GPU) -> the CPU oracle's rgb for the same rays
        (the checker, outside the timed region): PSNR of the HIP render against it goes into the row.
        Untimed warm-up: at least two calls AND 0.25 s of them - after the seconds of GPU idle that every row's CPU-oracle
        check leaves behind, the first ~2 ms of GPU work run at a reduced clock (measured round 3,
        profiles/r03/clock_ramp.md: the same 1024-ray launch 578 us as the first thing timed, 536 us after others), which a
        50 x 0.57 ms row would otherwise carry as 8 % of its time.  The timed region is >= 0.1 s."""
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            out = fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            one = max(time.perf_counter() - t0, 1e-5)
            for _ in range(int(0.25 / one)):
                fn()
            reps = max(reps, int(0.1 / one) + 1)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        tf = n_rays * flop_per_ray / dt / 1e12
        tfe = n_rays * (flop_per_ray if exec_per_ray is None else exec_per_ray) / dt / 1e12
        rows.append({"name": name, "ms": dt * 1e3, "rays_per_s": n_rays / dt, "algorithmic_tflops": tf, "executed_tflops": tfe,
                     "frac": tfe / PEAK_FP32_MFMA_TFLOPS, "reps": reps})
        if ref is not None:
            with torch.no_grad():
                want = ref()
            got = (out[0] if isinstance(out, (list, tuple)) else out[:, :3]).reshape(-1, 3)[:want.shape[0]].cpu()
            rows[-1]["psnr_vs_cpu_render_db"] = float(-10 * np.log10(max(float(((got - want) ** 2).mean()), 1e-20)))
            rows[-1]["psnr_rays"] = int(want.shape[0])

    st = build_scene("C2", dev, 0)
    kw, K8, c2w8 = st["kw"], st["K"], st["c2w"]
    K4, c2w4 = synth.lego_camera(400, 400)
    o, d = synth.pick_rays(400, 400, K4, c2w4, 1024, 1)
    r1 = (T(o), T(d))
    kw1 = dict(kw, N_importance=0, network_fine=None)
    sd_c, sd_f = (O.to_torch_sd(sd_) for sd_ in st["sds_np"])
    rb1 = O.make_ray_batch(torch.from_numpy(o), torch.from_numpy(d), 2., 6.)
    timeit("C1: 1024 rays x 64 coarse samples, one net", lambda: render.render(400, 400, K4, rays=r1, **kw1), 1024, 64 * FLOP_PER_ROW, 50,
           ref=lambda: O.render_rays(rb1, sd_c, None, N_SAMPLES, 0, white_bkgd=True)["rgb_map"], exec_per_ray=64 * FLOP_EXEC_PER_ROW + FLOP_EXEC_PER_RAY_PASS)
    # the north_star's own target line: lego (nerf/configs/lego.txt: half_res 400x400, N_rand = 1024, 64 + 128, two nets,
    # white_bkgd, use_viewdirs) - 1024 rays are exactly one wave per SIMD on 256 CUs: both launches run a single round
    timeit("north_star: lego 1024-ray batch x (64+128), two nets", lambda: render.render(400, 400, K4, rays=r1, **kw), 1024, st["flop_per_ray"], 50,
           ref=lambda: O.render_rays(rb1, sd_c, sd_f, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)["rgb_map"], exec_per_ray=st["exec_per_ray"])
    # use_viewdirs=False - the reference's argparse default (utils.py:43, model.py:59-60, 8-column rays nerf/run.py:152-157):
    # the C2 batch through two nets WITHOUT the view branch (output_ch = 5), on the fused pass's own variant; FLOPs per row
    # = 2 x (63*256 + 4*256^2 + 319*256 + 2*256^2 + 256*5) = 984 576
    from swnerf import model, embedder
    e_fn, in_ch = embedder.get_embedder(10, 3, 0)
    nv = []
    for seed, ab in ((20250321, 0.5), (20250322, 0.7)):          # opacity biases tuned with the CPU oracle (acc spans 0.2..1.0)
        m = model.vallina_NeRF(D=8, W=256, input_ch=in_ch, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
        sd_np = synth.noview_state_dict(seed, alpha_bias=ab)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        nv.append((m.to(dev).eval(), O.to_torch_sd(sd_np)))
    embed_fn, embeddirs_fn = e_fn, None
    q_nv = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
    o2, d2 = synth.pick_rays(800, 800, K8, c2w8, N_RAND, 2)
    r2 = (T(o2), T(d2))
    kw_nv = dict(kw, use_viewdirs=False, network_fn=nv[0][0], network_fine=nv[1][0], network_query_fn=q_nv)
    flop_nv = 2 * (63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256 + 256 * 5)

    def ref_nv():
        # the oracle's one-net render twice over: coarse net for the depths, fine net on them (nerf/run.py:394-413)
        rb = O.make_ray_batch(torch.from_numpy(o2[:256]), torch.from_numpy(d2[:256]), 2., 6.)[:, :8]
        mlp = lambda sd_: (lambda e: O.generic_mlp(sd_, e, 8, [4], 63, 0, False))
        return O.render_rays_two_nets_generic(rb, mlp(nv[0][1]), mlp(nv[1][1]), N_SAMPLES, N_IMPORTANCE, white_bkgd=True)["rgb_map"]
    timeit("C2 shape with use_viewdirs=False (the reference's default): 4096 x (64+128), two 8x256 nets without the view branch, fused",
           lambda: render.render(800, 800, K8, rays=r2, **kw_nv), N_RAND, (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * flop_nv, 10, ref=ref_nv)
    Kf, c2wf = synth.fern_camera()
    o, d = synth.pick_rays(378, 504, Kf, c2wf, N_RAND, 3)
    r3 = (T(o), T(d))
    kw3 = dict(kw, ndc=True, near=0., far=1., white_bkgd=False)
    rb3 = O.make_ray_batch(torch.from_numpy(o[:512]), torch.from_numpy(d[:512]), 0., 1., ndc=True, H=378, W=504, focal=float(Kf[0][0]))
    timeit("C3: fern-like NDC rays, 4096 x (64+128)", lambda: render.render(378, 504, Kf, rays=r3, **kw3), N_RAND, st["flop_per_ray"], 10,
           ref=lambda: O.render_rays(rb3, sd_c, sd_f, N_SAMPLES, N_IMPORTANCE, white_bkgd=False)["rgb_map"], exec_per_ray=st["exec_per_ray"])
    lo, hi = synth.shard_range(800 * 800, 8, 3)
    rr4 = parallel.frame_renderer(800, 800, K8, c2w8, kw, device=dev)
    o8, d8 = O.get_rays(800, 800, K8, c2w8)
    rb4 = O.make_ray_batch(o8.reshape(-1, 3)[lo:lo + 512], d8.reshape(-1, 3)[lo:lo + 512], 2., 6.)
    timeit("C4 shard: rank 3 of 8 of the 800x800 frame, 80 000 rays incl. get_rays", lambda: rr4(lo, hi - lo), hi - lo, st["flop_per_ray"], 3,
           ref=lambda: O.render_rays(rb4, sd_c, sd_f, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)["rgb_map"], exec_per_ray=st["exec_per_ray"])
    # the opt-in bf16x3 arithmetic (csrc/mlp_core_x3.h) on the same C4 shard: NOT the headline (that is fp32) - rays/s, the
    # PSNR against the same CPU oracle rays and against this library's fp32 render of the whole shard, and the bf16 MFMA
    # rate it sustains (3 MFMAs per product: 3 x the algorithmic FLOPs) against the 2.5 PFLOP/s dense bf16 peak
    with torch.no_grad():
        ref4 = rr4(lo, hi - lo)[:, :3].clone()
    for mode, what, share in (("bf16x3", "both passes", 1.0), ("bf16x3-fine", "fine pass only, the coarse pass that feeds the resampling stays fp32", 0.75)):
        prev = render.set_precision(mode)
        try:
            timeit(f"C4 shard, {mode} arithmetic (opt-in; 3 bf16 MFMAs per product, fp32 accumulate; {what})", lambda: rr4(lo, hi - lo), hi - lo,
                   st["flop_per_ray"], 3, ref=lambda: O.render_rays(rb4, sd_c, sd_f, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)["rgb_map"])
            with torch.no_grad():
                got4 = rr4(lo, hi - lo)[:, :3]
        finally:
            render.set_precision(prev)
        r = rows[-1]
        r["dtype"] = mode
        r["psnr_vs_fp32_pass_db"] = float(-10 * torch.log10(torch.mean((got4.double() - ref4.double()) ** 2)))
        if share == 1.0:
            r["bf16_mfma_tflops"] = 3 * r["algorithmic_tflops"]
            r["bf16_mfma_frac_of_2500"] = r["bf16_mfma_tflops"] / 2500.0
        r["frac"] = None                              # the fp32-MFMA roofline does not apply to these rows
    s5 = build_scene("C5", dev, 0)
    lo5, hi5 = synth.shard_range(400 * 400, 8, 3)
    sd_d = O.to_torch_sd(s5["sds_np"][0])
    o4, d4 = O.get_rays(400, 400, float(s5["K"][0][0]), s5["c2w"])
    for tv in (0.5, 0.0):
        rr5 = parallel.frame_renderer(400, 400, s5["K"], s5["c2w"], s5["kw"], frame_time=tv, device=dev)
        rb5 = O.make_ray_batch(o4.reshape(-1, 3)[lo5:lo5 + 256], d4.reshape(-1, 3)[lo5:lo5 + 256], 2., 6., frame_time=tv)
        timeit(f"C5 shard: D-NeRF rank 3 of 8 of the 400x400 frame, 20 000 rays, t={tv}", lambda: rr5(lo5, hi5 - lo5), hi5 - lo5,
               s5["flop_per_ray"] if tv else st["flop_per_ray"], 3,
               ref=lambda rb5=rb5: O.render_rays_dnerf(rb5, sd_d, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)["rgb_map"],
               exec_per_ray=s5["exec_per_ray"] if tv else st["exec_per_ray"])
        if tv:
            # the resampling is a discontinuous function of the coarse weights and gamma(x + dx) multiplies a last-bit difference of dx
            # by 2^9 in front of it: this figure moves by several dB with ANY last-bit change of the coarse pass (62.2 dB before the
            # per-ray gamma(t) tile, 56.4 after, on these 256 rays; 57.4 dB against the REFERENCE's own golden render, where the
            # reference against itself under a 2e-7 shift of dx reads 54.5 dB: tests/test_gpu_parity.py::test_render_rays_dnerf_golden).
            # The pass WITHOUT resampling holds dx 1e-6 / rgb 2e-5 against the oracle.
            rows[-1]["psnr_note"] = ("end-to-end D-NeRF PSNR is conditioning, not arithmetic: the reference against itself under a 2e-7 shift of dx "
                                     "reads 54.5 dB on the golden rays (tests/test_gpu_parity.py); without resampling dx agrees to 1e-6")
    # ... and the D-NeRF shard at t = 0.5 in bf16x3 (deformation + canonical net in one pass; the fp32 row is two rows up)
    rr5 = parallel.frame_renderer(400, 400, s5["K"], s5["c2w"], s5["kw"], frame_time=0.5, device=dev)
    rb5 = O.make_ray_batch(o4.reshape(-1, 3)[lo5:lo5 + 256], d4.reshape(-1, 3)[lo5:lo5 + 256], 2., 6., frame_time=0.5)
    with torch.no_grad():
        ref5 = rr5(lo5, hi5 - lo5)[:, :3].clone()
    for mode in ("bf16x3", "bf16x3-fine"):
        prev = render.set_precision(mode)
        try:
            timeit(f"C5 shard t=0.5, {mode} arithmetic (opt-in)", lambda: rr5(lo5, hi5 - lo5), hi5 - lo5, s5["flop_per_ray"], 3,
                   ref=lambda: O.render_rays_dnerf(rb5, sd_d, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)["rgb_map"])
            with torch.no_grad():
                got5 = rr5(lo5, hi5 - lo5)[:, :3]
        finally:
            render.set_precision(prev)
        r = rows[-1]
        r["dtype"] = mode
        r["psnr_vs_fp32_pass_db"] = float(-10 * torch.log10(torch.mean((got5.double() - ref5.double()) ** 2)))
        if mode == "bf16x3":
            r["bf16_mfma_tflops"] = 3 * r["algorithmic_tflops"]
            r["bf16_mfma_frac_of_2500"] = r["bf16_mfma_tflops"] / 2500.0
        r["frac"] = None
    # training step of the reference (nerf/run.py:684-708): render -> img2mse -> backward -> Adam; 3x the forward FLOPs
    nets = [kw["network_fn"], kw["network_fine"]]
    for m in nets:
        m.train()
    o, d = synth.pick_rays(800, 800, K8, c2w8, N_RAND, 2)
    rt = (T(o), T(d))
    target = torch.rand((N_RAND, 3), device=dev)
    opt = torch.optim.Adam([p for m in nets for p in m.parameters()], lr=5e-4, betas=(0.9, 0.999))
    kwt = dict(kw, perturb=1.)

    def train_step():
        rgb, disp, acc, extras = render.render(800, 800, K8, chunk=1024 * 32, rays=rt, **kwt)
        loss = torch.mean((rgb - target) ** 2) + torch.mean((extras['rgb0'] - target) ** 2)
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.reset_peak_memory_stats(dev)
    timeit("training step: 4096 rays x (64+128), two nets, mse(rgb)+mse(rgb0), backward, Adam (3x forward FLOPs)", train_step,
           N_RAND, 3 * st["flop_per_ray"], 5, grad=True, exec_per_ray=(N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * FLOP_EXEC_TRAIN_PER_ROW + 2 * FLOP_EXEC_PER_RAY_PASS)
    rows[-1]["peak_mem_gib"] = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    for m in nets:
        m.eval()
        for p in m.parameters():
            p.grad = None
    del opt
    # the same step for a use_viewdirs=False run (the reference's argparse default): fused training pass of the nets without
    # the view branch (swnerf_render_pass_train kind SWNERF_NET_NOVIEW + swnerf_render_pass_backward_noview)
    nets_nv = [nv[0][0], nv[1][0]]
    for m in nets_nv:
        m.train()
    opt_nv = torch.optim.Adam([p for m in nets_nv for p in m.parameters()], lr=5e-4, betas=(0.9, 0.999))
    kwt_nv = dict(kw_nv, perturb=1.)

    def train_step_nv():
        rgb, disp, acc, extras = render.render(800, 800, K8, chunk=1024 * 32, rays=r2, **kwt_nv)
        loss = torch.mean((rgb - target) ** 2) + torch.mean((extras['rgb0'] - target) ** 2)
        opt_nv.zero_grad()
        loss.backward()
        opt_nv.step()
    torch.cuda.reset_peak_memory_stats(dev)
    timeit("training step with use_viewdirs=False: 4096 rays x (64+128), two nets without the view branch, backward, Adam", train_step_nv,
           N_RAND, 3 * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE) * flop_nv, 5, grad=True)
    rows[-1]["peak_mem_gib"] = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    for m in nets_nv:
        m.eval()
        for p in m.parameters():
            p.grad = None
    del opt_nv
    # D-NeRF training step (d_nerf/run_dnerf.py:686-735, the shipped one-model configuration): the coarse pass runs under
    # no_grad and only feeds the resampling, the fine pass trains deformation + canonical net; image loss.  FLOPs:
    # coarse forward (64 rows) + 3 x fine (192 rows), each row through both nets
    dn = s5["kw"]["network_fn"]
    dn.train()
    o, d = synth.pick_rays(400, 400, s5["K"], s5["c2w"], N_RAND, 5)
    rd = (T(o), T(d))
    opt_d = torch.optim.Adam(dn.parameters(), lr=5e-4, betas=(0.9, 0.999))
    kwd = dict(s5["kw"], perturb=1.)
    focal = float(s5["K"][0][0])

    def train_step_dnerf():
        rgb, disp, acc, extras = render_dnerf.render(400, 400, focal, chunk=1024 * 32, rays=rd, frame_time=0.5, retraw=True, **kwd)
        loss = torch.mean((rgb - target) ** 2)
        opt_d.zero_grad()
        loss.backward()
        opt_d.step()
    torch.cuda.reset_peak_memory_stats(dev)
    timeit("D-NeRF training step: 4096 rays x (64+128), one DirectTemporalNeRF at t=0.5, mse(rgb), backward, Adam", train_step_dnerf,
           N_RAND, (N_SAMPLES + 3 * (N_SAMPLES + N_IMPORTANCE)) * (FLOP_PER_ROW + FLOP_PER_ROW_DEFORM), 4, grad=True,
           exec_per_ray=N_SAMPLES * (FLOP_EXEC_PER_ROW + FLOP_EXEC_PER_ROW_DEFORM) + 2 * (FLOP_EXEC_PER_RAY_PASS + FLOP_EXEC_PER_RAY_PASS_DEFORM)
           + (N_SAMPLES + N_IMPORTANCE) * (FLOP_EXEC_TRAIN_PER_ROW + 2 * 32768 + FLOP_EXEC_TRAIN_DEFORM_PER_ROW))   # + the 128 d gamma(x+dx) steps
    rows[-1]["peak_mem_gib"] = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    dn.eval()
    for p in dn.parameters():
        p.grad = None
    return rows


def worker(args):
    # hipcc (subprocesses) strictly BEFORE anything in this process initialises the GPU; under torchrun every rank gets
    # here, so the compile sits behind a file lock and only the first one in does the work
    import __graft_entry__
    __graft_entry__.compile_library_locked()
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: the launcher and the flag disagree")
    # SWNERF_BENCH_REHEARSAL=1: run the N>1 control flow on a box with fewer GPUs than ranks (ranks share
    # cards, gloo instead of RCCL, pixels staged through the host for the gather).  Not a measurement.
    rehearsal = os.environ.get("SWNERF_BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()                     # counts only: no context yet
    if ndev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    if local_rank >= ndev and not rehearsal:
        raise SystemExit(f"[bench] rank {rank}: local rank {local_rank} but only {ndev} GPU(s) visible; refusing to share a card "
                         f"under a {world}-GPU label")
    dev_index = local_rank % ndev if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    collective = world > 1 or args.collective == "always"
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:              # only a bare N = 1 run gets here: nobody else needs the port
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        tmo = datetime.timedelta(seconds=600)
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=tmo, device_id=dev)
    pr = torch.cuda.get_device_properties(dev)
    pci = ":".join(f"{getattr(pr, k):02x}" for k in ("pci_domain_id", "pci_bus_id", "pci_device_id") if isinstance(getattr(pr, k, None), int))
    print(f"[bench] rank {rank}/{world}: cuda:{dev_index} = {pr.name} ({getattr(pr, 'gcnArchName', '?')}, {pr.multi_processor_count} CUs, "
          f"{pr.total_memory / 2 ** 30:.0f} GiB, pci {pci or '?'}, uuid {getattr(pr, 'uuid', '?')}); "
          f"collective: {dist.get_backend() if collective else 'none'}", file=sys.stderr, flush=True)

    __graft_entry__.check_library()                      # symbols only; starts no program
    if collective:
        dist.barrier()
    from swnerf import synth, render, parallel
    render.set_precision(args.precision)
    x3 = args.precision != "fp32"

    cfg = args.config
    sc = build_scene(cfg, dev, rank)
    H, W, K, c2w, kw = sc["H"], sc["W"], sc["K"], sc["c2w"], sc["kw"]
    S_FINE = N_SAMPLES + N_IMPORTANCE

    fine_events = []

    def hook(phase, n_rays, n_samples):
        if n_samples == S_FINE and hook.on:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))     # the stream the kernel is launched on
            fine_events.append(ev)
    hook.on = False
    render.PASS_HOOK = hook

    pending = []                                         # all-gathers in flight: (work, out)

    def gather(px):
        """The step's collective, issued asynchronously: the all-gather of this step's pixels is enqueued behind the render
        that produced them and runs on RCCL's stream under the NEXT step's first launch (one collective in flight at most;
        finish() drains before the clock stops, so every step's gather is inside the timed region)."""
        if not collective:
            return px
        if rehearsal:
            return parallel.gather_pixels(px.cpu(), force=True).to(dev)
        while len(pending) >= 1:
            pending.pop(0)[0].wait()
        out, work = parallel.gather_pixels(px, force=True, async_op=True)    # all_gather_into_tensor on device tensors, also at world 1
        pending.append((work, out))
        return out

    def finish():
        while pending:
            pending.pop(0)[0].wait()

    train = cfg == "train"
    if cfg in ("C2", "train"):
        o_np, d_np = synth.pick_rays(H, W, K, c2w, N_RAND, seed=2 + rank)
        rays_o, rays_d = torch.from_numpy(o_np).to(dev), torch.from_numpy(d_np).to(dev)
        n_local, rays_per_step, scaling = N_RAND, world * N_RAND, "weak"

        def local():                                     # this rank's part of a step: no collective
            rgb, disp, acc, _ = render.render(H, W, K, chunk=1024 * 32, rays=(rays_o, rays_d), **kw)
            return torch.cat([rgb, disp[:, None], acc[:, None]], -1)
        if train:
            # the reference's training step; every rank draws its own batch (seed 2 + rank), gradients are averaged
            nets_t = [kw["network_fn"], kw["network_fine"]]
            for m in nets_t:
                m.train()
            tgt = torch.rand((N_RAND, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(11 + rank))
            opt = torch.optim.Adam([p_ for m in nets_t for p_ in m.parameters()], lr=5e-4, betas=(0.9, 0.999))
            bucket = parallel.GradBucket(nets_t, force=collective)
            kwt = dict(kw, perturb=1.)

            def local():
                with torch.enable_grad():
                    rgb, disp, acc, ex = render.render(H, W, K, chunk=1024 * 32, rays=(rays_o, rays_d), **kwt)
                    loss = torch.mean((rgb - tgt) ** 2) + torch.mean((ex['rgb0'] - tgt) ** 2)
                    bucket.zero()
                    loss.backward()
                bucket.wait()                            # the all-reduces were issued from autograd's hooks; no-op without a group
                opt.step()
                return torch.cat([rgb.detach(), disp.detach()[:, None], acc.detach()[:, None]], -1)
    else:
        lo, hi = synth.shard_range(H * W, world, rank)
        n_local, rays_per_step, scaling = hi - lo, H * W, "strong"
        if (H * W) % world:
            raise SystemExit(f"[bench] {cfg}: {H * W} rays do not split evenly over {world} ranks")
        render_range = parallel.frame_renderer(H, W, K, c2w, kw, frame_time=sc["frame_time"], device=dev)

        def local():
            return render_range(lo, hi - lo)

    def step():
        return local() if train else gather(local())        # the training step's collective is the gradient all-reduce inside local()

    def fence():
        finish()
        torch.cuda.synchronize(dev)
        if collective:
            dist.barrier()
        torch.cuda.synchronize(dev)

    PREWARM_S = 0.25
    with torch.no_grad():
        # clock pre-warm (untimed, before the W warm-up steps, reported as config.clock_prewarm_s): the scene set-up above is
        # seconds of GPU idle, and the first milliseconds of GPU work after an idle spell run at a reduced clock
        # (profiles/r03/clock_ramp.md); W = 5 steps of 8 ms do not reliably cover that
        # - WITHOUT the collective: the number of pre-warm renders differs from rank to rank, collectives must not
        t_pre = time.perf_counter()
        if train:                                        # its local() carries the gradient all-reduce: a FIXED number of steps on every rank
            for _ in range(8):
                local()
            torch.cuda.synchronize(dev)
        while not train and time.perf_counter() - t_pre < PREWARM_S:
            local()
            torch.cuda.synchronize(dev)
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
    if collective:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert out.shape == ((n_local if train else rays_per_step), 5) and bool(torch.isfinite(out[:, :3]).all())

    ms_kernel = [fine_events[i].elapsed_time(fine_events[i + 1]) for i in range(0, len(fine_events), 2)]
    fine_ms = float(np.mean(ms_kernel)) if ms_kernel else float("nan")
    if train:
        args.no_cpu_baseline = args.no_extra = True
    fine_flop = n_local * (S_FINE * sc["exec_per_fine_row"] + sc["exec_per_ray_pass"])   # MFMA FLOPs the launch EXECUTES (fold + per-ray gamma(d) [+ gamma(t)])
    fine_flop_alg = n_local * S_FINE * sc["flop_per_fine_row"]          # ... the reference's algorithm would
    achieved = fine_flop / (fine_ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath) and cfg == "C2":
        # HBM/fabric bytes per fine-pass launch from the PMC passes of the round named in the file (tools/profile_r03.sh ->
        # tools/make_roofline_traffic.py); a counter figure cannot be collected inside this run
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("fine_pass_hbm_bytes_per_launch")
            traffic_src = f"round {tj.get('round')}: {tj.get('source')}"
        except Exception:
            traffic = None

    workloads = {
        "C2": "C2: lego-like 800x800 camera, N_rand=4096 rays/GPU/step, 64 coarse + 128 fine samples, coarse+fine 8x256 nets "
              "(use_viewdirs), white_bkgd, perturb=0; render() forward (get rays resident -> ray batch -> coarse pass -> "
              "resample -> fine pass)",
        "C4": "C4: lego-like 800x800 full-image render_only (640000 rays/step over all GPUs), 64+128, coarse+fine 8x256 nets; "
              "each rank: get_rays on its contiguous row range -> ray batch -> coarse pass -> resample -> fine pass",
        "C5": "C5: D-NeRF (bouncingballs-like) 400x400 full-image render (160000 rays/step over all GPUs) at t=0.5, 64+128, "
              "one DirectTemporalNeRF (deformation + canonical 8x256 net per sample); each rank renders its contiguous row range",
        "train": "TRAINING step of the reference (nerf/run.py:684-708) on the C2 shape: render(4096 rays/GPU, 64+128, coarse+fine 8x256 "
                 "nets, perturb=1) -> mse(rgb)+mse(rgb0) -> backward through the fused passes -> Adam; NOT the BASELINE metric",
    }
    result = {
        "metric": "rays/sec (64+128 samples/ray)" + (", training step" if train else ""), "value": rays_per_step * args.steps / dt, "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32" if not x3 else args.precision,
        "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, gloo; not a measurement)" if rehearsal else ""),
        "config": {"workload": workloads[cfg] + (("; + RCCL all-reduce of the gradients" if train else "; + RCCL all-gather of [rgb,disp,acc]") if collective else ""),
                   "collective": ((f"{dist.get_backend()} all_reduce of the gradient buckets ({bucket.nbytes()} B in {len(bucket.items)} in-place buckets, "
                                   f"async from autograd hooks) per step + all_reduce(MAX) of the time, world {world}" if train else
                                   f"{dist.get_backend()} all_gather_into_tensor of the [n,5] pixels per step + all_reduce(MAX) of the time, world {world}")
                                  if collective else "none (N = 1)"),
                   "rays_per_step_per_gpu": n_local, "n_samples": N_SAMPLES, "n_importance": N_IMPORTANCE, "clock_prewarm_s": PREWARM_S,
                   "parallelism": f"ray-sharded dp{world}"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": f"{sc['kernel']} fine pass ({n_local} rays x {S_FINE} samples)" + (", TRAIN forward variant" if train else ""),
                     "ms_per_launch": fine_ms, "flop_per_launch": fine_flop,
                     "flop_accounting": "executed: feature_linear (65 536 MACs/row, no activation) is folded into views_linears.0 at pack "
                                        "time and the view layer's gamma(d) columns are evaluated once per ray (131 072 MACs) instead of per "
                                        "row: 523 776 MACs/row run (+491 776 for the deformation net, whose gamma(t) columns are per-ray too: +262 144 MACs per ray and pass); algorithmic_* = the reference's 593 408",
                     "algorithmic_flop_per_launch": fine_flop_alg, "algorithmic_tflops": fine_flop_alg / (fine_ms * 1e-3) / 1e12,
                     "step_frac": rays_per_step / world * ((S_FINE + N_SAMPLES) * (FLOP_EXEC_TRAIN_PER_ROW if train else sc["exec_per_fine_row"])
                                                           + 2 * sc["exec_per_ray_pass"]) / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
    }
    if x3:
        # the fine pass runs 3 bf16 MFMAs per product: price the matrix work it really does against the dense bf16 peak
        r = result["roofline"]
        r.update({"achieved": 3 * achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "frac": 3 * achieved / PEAK_BF16_MFMA_TFLOPS, "traffic": None,
                  "flop_per_launch": 3 * fine_flop, "step_frac": None,
                  "note": "bf16x3: achieved = 3 x the algorithmic FLOPs of the fine pass / its launch time; NOT the BASELINE metric's dtype"})
        result["metric"] += f" [{args.precision} arithmetic, opt-in]"
        args.no_extra = True
    assert result["n_gpus"] == args.gpus

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import nerf_oracle as O
        torch.set_num_threads(host_cores())
        cores = torch.get_num_threads()
        if cfg == "C2":
            o_cpu, d_cpu, gpu_rows = torch.from_numpy(o_np), torch.from_numpy(d_np), out[:N_RAND, :3]
            what = "the same 4096-ray batch"
        else:   # a bounded sample of the frame: 4096 pixels spread evenly over it
            sel = np.linspace(0, H * W - 1, N_RAND).astype(np.int64)
            o_all, d_all = O.get_rays(H, W, K, c2w)
            o_cpu, d_cpu, gpu_rows = o_all.reshape(-1, 3)[sel], d_all.reshape(-1, 3)[sel], out[torch.from_numpy(sel).to(dev), :3]
            what = f"4096 pixels spread evenly over the {H}x{W} frame"
        rb = O.make_ray_batch(o_cpu, d_cpu, 2., 6., frame_time=sc["frame_time"])
        if cfg == "C5":
            sd = O.to_torch_sd(sc["sds_np"][0])
            run = lambda r: O.render_rays_dnerf(r, sd, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)
        else:
            sd_c, sd_f = (O.to_torch_sd(s) for s in sc["sds_np"])
            run = lambda r: O.render_rays(r, sd_c, sd_f, N_SAMPLES, N_IMPORTANCE, white_bkgd=True)
        with torch.no_grad():
            run(rb[:256])                                                                        # warm-up
            # bounded sample: the whole batch once (needed for the PSNR), then repeat up to ~15 s / 5 reps
            reps, t0 = 0, time.perf_counter()
            while reps < 1 or (time.perf_counter() - t0 < 15.0 and reps < 5):
                ref = run(rb)
                reps += 1
            cdt = time.perf_counter() - t0
        mse = float(((gpu_rows.cpu() - ref["rgb_map"]) ** 2).mean())
        result["cpu_baseline"] = {"value": reps * N_RAND / cdt, "unit": "rays/s", "cores": cores, "kind": "port",
                                  "sample": f"{reps} x {what} (64+128), oracle/nerf_oracle.py, "
                                            f"torch CPU {torch.__version__}, no_grad"}
        result["psnr_vs_cpu_render_db"] = float(-10 * np.log10(max(mse, 1e-20)))
    if rank == 0 and world == 1 and not args.no_extra:
        render.PASS_HOOK = None
        result["extra"] = {"note": "measured after and outside the headline's timed region, same process and GPU; wall clock per "
                                   "call incl. Python; frac = EXECUTED MLP FLOPs / time / 157.3 TFLOP/s (executed_tflops; feature_linear is folded into the view "
                                   "layer, training rows count forward + dX chain + dW as run), algorithmic_tflops = the reference's FLOP count "
                                   "(3 x forward for training) / time; psnr_vs_cpu_render_db = the "
                                   "HIP render against the CPU oracle's render of the first psnr_rays rays of that config",
                           "configs": extra_configs(dev)}
        # the HBM / VALU-issue bound satellites of the path (get_rays, ray-batch pack, Embedder, raw2outputs forward and
        # backward, sample_pdf + sort as standalone ops), at sizes where they are not launch bound: us, algorithmic bytes, GB/s,
        # fraction of the 8 TB/s spec and of the 6.29 TB/s a float4 copy reaches (tools/bench_satellites.py; PMC evidence of the
        # same launches: profiles/r04/satellites*.md)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_satellites
        torch.cuda.empty_cache()
        result["extra"]["satellites"] = bench_satellites.run_satellites(dev, reps=10, quiet=True)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args))
    worker(args)


if __name__ == "__main__":
    main()
