"""bench.py --gpus N without a launcher must start N rank processes itself and report n_gpus == N (round-1 VERDICT:
it silently measured one GPU).  On the one-GPU test box the ranks share the card over gloo
(SWNERF_BENCH_REHEARSAL=1: control flow only, not a measurement): launcher -> 2 fresh ranks -> each renders ITS half
of the 800x800 frame with the HIP renderer (get_rays on the range, fused passes) -> all-gather -> one JSON line.

This module sorts first on purpose: it must run before anything in the pytest process has initialised the GPU (a
process that has may not start other programs on this pool)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks():
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: starting programs from it is not allowed on this pool")
    env = dict(os.environ, SWNERF_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C4", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["rays_per_step_per_gpu"] == 320000
    assert j["config"]["parallelism"] == "ray-sharded dp2" and "REHEARSAL" in j["data"]
    assert j["value"] > 0 and j["roofline"]["frac"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_rccl_leg_at_world_1():
    """The code path the driver's N > 1 runs hit first, on the one GPU this box has: `bench.py --collective always` joins an
    RCCL group of ONE rank (init_process_group("nccl", device_id=...)), every step ends with all_gather_into_tensor on the
    DEVICE tensor of the pixels and the timing goes through all_reduce(MAX).  The number must agree with the
    no-collective N = 1 run (the gather of 80 KB is latency only) - that is also the "N = 1 of SCALE agrees with BENCH"
    check.  Both runs are fresh child processes, started before this process has touched the GPU."""
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: starting programs from it is not allowed on this pool")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SWNERF_BENCH_REHEARSAL")}
    lines = {}
    for mode in ("always", "auto"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--collective", mode, "--steps", "30", "--warmup", "5",
                            "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True, timeout=400)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        if mode == "always":
            assert "collective: nccl" in r.stderr and "rank 0/1: cuda:0 = " in r.stderr, r.stderr[-2000:]
    a, b = lines["always"], lines["auto"]
    assert a["n_gpus"] == 1 and "nccl all_gather_into_tensor" in a["config"]["collective"] and "RCCL all-gather" in a["config"]["workload"]
    assert b["config"]["collective"].startswith("none")
    print(f"[rccl world 1] with collective {a['value']:.0f} rays/s ({a['ms_per_step']:.3f} ms/step), without {b['value']:.0f} rays/s "
          f"({b['ms_per_step']:.3f} ms/step): ratio {a['value'] / b['value']:.4f}")
    # functional assertions above; the throughput comparison is informational with a loose gate (clock state after idle moves
    # two separate 30-step child runs by several per cent, profiles/r03/clock_ramp.md - measured ratios 0.996-0.998)
    assert 0.9 < a["value"] / b["value"] < 1.1, (a["value"], b["value"])
    assert abs(a["roofline"]["frac"] - b["roofline"]["frac"]) < 0.05


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_training_collective_on_rccl_at_world_1():
    """The data-parallel TRAINING collective on RCCL, on the one GPU this box has: `bench.py --config train --collective
    always` runs the reference's training step (nerf/run.py:684-708) with the gradients in swnerf.parallel.GradBucket - one
    in-place ncclAllReduce per net in a group of one rank, issued async from autograd's post-accumulate hooks, waited for in
    front of optimizer.step() - the code path the driver's N > 1 training run would take.  Compared with the same step
    without a process group (fresh child processes, started before this one touches the GPU)."""
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: starting programs from it is not allowed on this pool")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SWNERF_BENCH_REHEARSAL")}
    lines = {}
    for mode in ("always", "auto"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "train", "--collective", mode, "--steps", "10",
                            "--warmup", "2"], env=env, capture_output=True, text=True, timeout=400)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        if mode == "always":
            assert "collective: nccl" in r.stderr, r.stderr[-2000:]
    a, b = lines["always"], lines["auto"]
    assert a["n_gpus"] == 1 and "training step" in a["metric"] and "nccl all_reduce of the gradient buckets" in a["config"]["collective"]
    assert "2 in-place buckets" in a["config"]["collective"] and "RCCL all-reduce of the gradients" in a["config"]["workload"]
    assert b["config"]["collective"].startswith("none") and "cpu_baseline" not in a and "extra" not in a
    print(f"[rccl world 1, training] with the gradient all-reduce {a['value']:.0f} rays/s ({a['ms_per_step']:.3f} ms/step), without "
          f"{b['value']:.0f} rays/s ({b['ms_per_step']:.3f} ms/step): ratio {a['value'] / b['value']:.4f}")
    assert 0.9 < a["value"] / b["value"] < 1.1, (a["value"], b["value"])


def test_bench_refuses_more_gpus_than_visible():
    """No launcher, --gpus 2, fewer GPUs than ranks and no rehearsal flag: exit non-zero, no JSON (never a 1-GPU number
    under an N-GPU label).  Runs anywhere: the launcher parent touches no GPU."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "SWNERF_BENCH_REHEARSAL")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "{" not in r.stdout and "refusing" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_c_abi_from_plain_cpp_matches_python_mirror(tmp_path):
    """examples/cabi_demo.cpp drives libswnerf_hip.so with hipMalloc'd buffers only (no Python, no torch): get_rays ->
    pack_ray_batch -> coarse pass + resampling -> fine pass.  The same render through the reference-shaped Python
    surface (render.render with the create_nerf kwargs) on the weights the demo dumped must agree BIT FOR BIT: both
    are the same kernels behind the same C ABI, the Python side adds nothing to the arithmetic."""
    import numpy as np
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: starting programs from it is not allowed on this pool")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.compile_library()
    libdir = os.path.join(ROOT, "sw-nerf_amd", "swnerf")
    exe, dump = str(tmp_path / "cabi_demo"), str(tmp_path / "dump.bin")
    c = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "cabi_demo.cpp"), "-L", libdir, "-lswnerf_hip", f"-Wl,-rpath,{libdir}", "-o", exe],
                       capture_output=True, text=True, timeout=300)
    assert c.returncode == 0, c.stderr[-3000:]
    r = subprocess.run([exe, dump], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "cabi_demo: 256 rays" in r.stdout

    sys.path.insert(0, os.path.join(ROOT, "sw-nerf_amd"))
    import swnerf.embedder, swnerf.model, swnerf.render   # noqa
    import swnerf as sw
    blob = np.fromfile(dump, dtype=np.float32)
    names = list(sw.model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True).state_dict().keys())
    dev, pos, nets = torch.device("cuda:0"), 0, []
    for _ in range(2):
        m = sw.model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        sd = {}
        for k in names:
            n = m.state_dict()[k].numel()
            sd[k] = torch.from_numpy(blob[pos:pos + n].copy()).reshape(m.state_dict()[k].shape)
            pos += n
        m.load_state_dict(sd)
        nets.append(m.to(dev).eval())
    N = 256
    rgb0, rgb, acc = (blob[pos:pos + 3 * N].reshape(16, 16, 3), blob[pos + 3 * N:pos + 6 * N].reshape(16, 16, 3),
                      blob[pos + 6 * N:pos + 7 * N].reshape(16, 16))
    assert pos + 7 * N == blob.size
    embed_fn, _ = sw.embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = sw.embedder.get_embedder(4, 3, 0)
    query = lambda inputs, viewdirs, network_fn: sw.render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,  # noqa: E731
                                                                       embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    import math
    focal = 0.5 * 16 / math.tan(0.5 * 0.6911112070083618)
    K = np.array([[focal, 0, 8.0], [0, focal, 8.0], [0, 0, 1]])
    c2w = torch.tensor([[1., 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4]], device=dev)
    with torch.no_grad():
        out = sw.render.render(16, 16, K, chunk=1024 * 32, c2w=c2w, ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets[0],
                               network_query_fn=query, N_samples=64, N_importance=128, network_fine=nets[1], white_bkgd=True,
                               perturb=0., raw_noise_std=0.)
    assert np.array_equal(out[0].cpu().numpy(), rgb), np.abs(out[0].cpu().numpy() - rgb).max()
    assert np.array_equal(out[2].cpu().numpy(), acc)
    assert np.array_equal(out[3]["rgb0"].cpu().numpy(), rgb0)
    # a fog of varying density: every ray is absorbed by the far plane's 1e10-wide last bin (acc == 1 up to rounding) but the
    # colour is set along the way and differs between pixels and between the coarse and the fine net
    assert acc.min() > 0.99 and rgb.std() > 1e-3 and 0.05 < rgb.mean() < 0.95 and np.abs(rgb - rgb0).max() > 1e-3, (rgb.mean(), rgb.std())
