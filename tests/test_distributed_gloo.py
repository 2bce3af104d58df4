"""world_size-2 gloo test of the ray-sharded path: contiguous shards, one
all-gather of pixels, every rank ends with the full image.  The per-rank renderer is the CPU
oracle here (no GPU in this container); on the GPU box the same parallel.* code runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "sw-nerf_amd"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from swnerf import parallel, synth
    from oracle import nerf_oracle as O
    H, W = 9, 14                                   # 126 rays: ragged over 2 (63/63) and over the 4 below
    K, c2w = synth.lego_camera(H, W)
    sd = O.to_torch_sd(synth.nerf_state_dict(*synth.NET_COARSE[:1], alpha_bias=synth.NET_COARSE[1]))
    o, d = O.get_rays(H, W, K, c2w)
    rb = O.make_ray_batch(o, d, 2., 6.)

    def render_range(lo, n):
        with torch.no_grad():
            r = O.render_rays(rb[lo:lo + n], sd, None, 16, 0, white_bkgd=True)
        return torch.cat([r["rgb_map"], r["disp_map"][:, None], r["acc_map"][:, None]], -1)

    img = parallel.render_image_sharded(render_range, H, W)
    # equal shards -> the single-collective path; ragged shards -> pad/trim path
    a = torch.full((5, 3), float(rank))
    same = parallel.gather_pixels(a)
    same_async, work = parallel.gather_pixels(a, async_op=True)
    work.wait()
    assert torch.equal(same, same_async)
    rag = parallel.gather_pixels(torch.full((3 + rank, 2), float(rank)), counts=[3 + r for r in range(world)])
    # data-parallel training: one flat all-reduce of the gradients (each rank saw half of the rays)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    for k, p_ in enumerate(lin.parameters()):
        p_.grad = torch.full_like(p_, float(rank + 1) * (k + 1))
    parallel.allreduce_gradients([lin, None])
    grads_ok = all(bool(torch.all(p_.grad == 1.5 * (k + 1))) for k, p_ in enumerate(lin.parameters()))
    # a rank WITHOUT a gradient for some parameters (D-NeRF: a batch at frame_time == 0 takes the zero_canonical branch
    # and leaves `_time.*` without .grad, model.py:143-145): same element count on every rank, zeros where there is none
    lin2 = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2), torch.nn.Linear(2, 2))
    ps = list(lin2.parameters())
    for k, p_ in enumerate(ps):
        p_.grad = None
    for k in (0, 1):
        ps[k].grad = torch.full_like(ps[k], 2.0 * (rank + 1))            # layer 0: both ranks
    if rank == 1:
        for k in (2, 3):
            ps[k].grad = torch.full_like(ps[k], 8.0)                     # layer 1: rank 1 only; layer 2: nobody
    parallel.allreduce_gradients([lin2])
    grads_ok = grads_ok and all(bool(torch.all(ps[k].grad == 3.0)) for k in (0, 1)) \
        and all(ps[k].grad is not None and bool(torch.all(ps[k].grad == 4.0)) for k in (2, 3)) \
        and all(ps[k].grad is None for k in (4, 5))
    # the copy-free form: gradients live in one bucket per module, the all-reduce of a module starts from the hook of its
    # last accumulated parameter (async), wait() averages; a module whose backward never ran is reduced from wait()
    torch.manual_seed(7)
    ma = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 2))
    mb = torch.nn.Linear(3, 3)
    bucket = parallel.GradBucket([ma, None, mb])
    x = torch.full((4, 6), float(rank + 1))
    for step in range(2):                                    # twice: zero() must re-arm the hooks and keep the views
        bucket.zero()
        ma(x).sum().backward()                               # mb gets no gradient at all in this step
        launched_by_hook = bucket.items[0]["launched"] and not bucket.items[1]["launched"]
        bucket.wait()
        want = {}
        for r in range(world):
            mr = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 2))
            mr.load_state_dict(ma.state_dict())
            mr(torch.full((4, 6), float(r + 1))).sum().backward()
            for k, p_ in enumerate(mr.parameters()):
                want[k] = want.get(k, 0) + p_.grad / world
        grads_ok = grads_ok and launched_by_hook and all(torch.allclose(p_.grad, want[k], atol=1e-6) for k, p_ in enumerate(ma.parameters())) \
            and all(bool(torch.all(p_.grad == 0)) for p_ in mb.parameters()) \
            and all(p_.grad.data_ptr() >= bucket.items[0]["flat"].data_ptr() for p_ in ma.parameters())
    # the dense-grid query for mesh extraction shards the same way: each rank queries its contiguous
    # shard of the R^3 points, one all-gather returns the field; the per-point query is the CPU oracle here
    import swnerf.mesh as mesh
    sdm = O.to_torch_sd(synth.nerf_state_dict(*synth.NET_FINE[:1], alpha_bias=synth.NET_FINE[1]))
    qf = lambda p, dirs_: torch.cat([torch.stack([O.query_points(sdm, p, dirs_[v:v + 1].expand(p.shape[0], 3))[:, :3] for v in range(dirs_.shape[0])], 0).mean(0),
                                     O.query_points(sdm, p, dirs_[:1].expand(p.shape[0], 3))[:, 3:4]], -1)
    bounds = [(-1., 1.), (-1., 2.), (-4., 2.)]
    dens_s, col_s, _ = mesh.sample_grid(bounds, 5, None, num_views=3, batch_size=40, query=qf)              # 125 points: 63 / 62 over 2 ranks
    dens_1, col_1, _ = mesh.sample_grid(bounds, 5, None, num_views=3, batch_size=1000, query=qf, sharded=False)
    mesh_ok = bool(np.allclose(dens_s, dens_1, atol=1e-5) and np.allclose(col_s, col_1, atol=1e-5) and dens_s.shape == (5, 5, 5))
    # the bench's timing reduction: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((img.numpy(), render_range(0, H * W).reshape(H, W, 5).numpy(), same.numpy(), rag.numpy(), float(t), grads_ok and mesh_ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_render_gloo_world2():
    world, port = 2, 29571
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    img, ref, same, rag, tmax, grads_ok = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert img.shape == (9, 14, 5)
    np.testing.assert_allclose(img, ref, atol=2e-6, equal_nan=True)      # (sgemm blocking differs with the row count)
    assert np.array_equal(same[:5], np.zeros((5, 3))) and np.array_equal(same[5:], np.ones((5, 3)))
    assert rag.shape == (7, 2) and np.array_equal(rag[:3], np.zeros((3, 2))) and np.array_equal(rag[3:], np.ones((4, 2)))
    assert tmax == 2.0 and grads_ok


def test_gather_pixels_force_runs_the_collective_at_world_1():
    """bench.py --collective always: a group of ONE rank still goes through all_gather_into_tensor (gloo here, RCCL on
    the GPU box) instead of the early return - and without a process group `force` is an error, not a silent no-op."""
    for p in (ROOT, os.path.join(ROOT, "sw-nerf_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from swnerf import parallel
    a = torch.arange(15, dtype=torch.float32).reshape(5, 3)
    assert parallel.gather_pixels(a) is a
    with pytest.raises(RuntimeError):
        parallel.gather_pixels(a, force=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29573")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        calls = []
        real = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda out, inp, group=None, async_op=False: (calls.append(1), real(out, inp, group=group, async_op=async_op))[1]
        try:
            assert parallel.gather_pixels(a) is a and not calls                 # default: no collective for one rank
            g = parallel.gather_pixels(a, force=True)
        finally:
            dist.all_gather_into_tensor = real
        assert calls == [1] and g is not a and torch.equal(g, a)
        g2, work = parallel.gather_pixels(a, force=True, async_op=True)          # bench.py's form: wait before reading
        work.wait()
        assert torch.equal(g2, a) and parallel.gather_pixels(a, async_op=True) == (a, None)
    finally:
        dist.destroy_process_group()


def test_gradient_collectives_force_at_world_1():
    """bench.py --config train --collective always: GradBucket / allreduce_gradients run their all_reduce in a group of one
    rank (the code path of an N-GPU training job on a one-GPU box); without `force` a single rank does no collective, without
    a process group `force` is an error."""
    for p in (ROOT, os.path.join(ROOT, "sw-nerf_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from swnerf import parallel
    lin = torch.nn.Linear(3, 2)
    lin.weight.grad, lin.bias.grad = torch.ones_like(lin.weight), torch.ones_like(lin.bias)
    with pytest.raises(RuntimeError):
        parallel.allreduce_gradients([lin], force=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29575")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, op=dist.ReduceOp.SUM, group=None, async_op=False: (calls.append(t.numel()), real(t, op=op, group=group, async_op=async_op))[1]
        try:
            parallel.allreduce_gradients([lin])
            assert not calls
            parallel.allreduce_gradients([lin], force=True)
            assert calls == [6 + 2 + 2]                              # weights + bias + one flag per parameter
            assert bool(torch.all(lin.weight.grad == 1)) and bool(torch.all(lin.bias.grad == 1))
            calls.clear()
            m = torch.nn.Linear(3, 2)
            b0 = parallel.GradBucket([m])                            # no force, one rank: a gradient arena only
            b0.zero(); m(torch.ones(1, 3)).sum().backward(); b0.wait()
            assert not calls and bool(torch.all(m.bias.grad == 1))
            m2 = torch.nn.Linear(3, 2)
            b1 = parallel.GradBucket([m2], force=True)
            b1.zero(); m2(torch.ones(1, 3)).sum().backward(); b1.wait()
            assert calls == [8 + 4] and bool(torch.all(m2.weight.grad == 1))     # (6 -> 8, 2 -> 4: views padded to 16 bytes)
            m2.weight.grad = None
            with pytest.raises(RuntimeError):
                b1.zero()
        finally:
            dist.all_reduce = real
    finally:
        dist.destroy_process_group()


def test_launcher_counts_gpus_from_sysfs_without_torch(tmp_path, monkeypatch):
    """bench.py's launcher parent counts GPUs from the KFD topology (+ accessible render nodes + *_VISIBLE_DEVICES) so that
    it never imports torch or touches HIP before it starts the rank processes."""
    sys.path.insert(0, ROOT)
    import bench
    nodes, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir()
    for i, (simd, minor) in enumerate([(0, -1), (1024, 128), (1024, 129), (1024, 130)]):     # a CPU node and three GPUs
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\ndrm_render_minor {minor}\n")
    for m in (128, 129):                                                                     # the container may open two of them
        (dri / f"renderD{m}").write_text("")
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpu_count(str(nodes), str(dri)) == 2
    assert bench.visible_gpu_count(str(nodes), str(tmp_path / "no_dri")) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count(str(nodes), str(dri)) == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count(str(nodes), str(dri)) == 0
    assert bench.visible_gpu_count(str(tmp_path / "nothing"), str(dri)) == 0
    import inspect
    src = inspect.getsource(bench.launch) + inspect.getsource(bench.visible_gpu_count)
    assert "import torch" not in src and "torch." not in src
