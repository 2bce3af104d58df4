"""Ray-sharded data parallelism.  The reference has no distributed
code; rays are independent, so the full-image render is split into contiguous row-major
shards, one process per GPU, and the only exchange is ONE all-gather of the rendered
pixels [rgb(3), disp, acc] (1.6 MB per rank for an 800x800 frame on 8 GPUs) - RCCL over
xGMI when the process group's backend is "nccl", gloo in the CPU tests."""
import os

import torch
import torch.distributed as dist

from .synth import shard_range


def gather_pixels(local, counts=None, group=None, force=False, async_op=False):
    """all-gather of per-rank [n_local, C] pixel blocks into [sum n, C] on every rank.
    Equal shards use a single all_gather_into_tensor (one RCCL call); ragged shards pad to the
    largest and trim.  A group of one rank returns `local` without a collective unless `force` (bench.py --collective
    always: the RCCL call of the N > 1 runs, exercised on a one-GPU box).
    async_op (equal shards only): returns (out, work) with the collective enqueued behind the producer of `local` on the
    backend's own stream - the caller renders on and calls work.wait() before it reads `out` (work is None when no
    collective was needed): the 1.6 MB gather of a frame then runs under the next frame's first launch instead of in
    front of it."""
    if not (dist.is_available() and dist.is_initialized()):
        if force:
            raise RuntimeError("swnerf.parallel.gather_pixels(force=True) needs an initialised process group")
        return (local, None) if async_op else local
    if dist.get_world_size(group) == 1 and not force:
        return (local, None) if async_op else local
    world = dist.get_world_size(group)
    n, c = local.shape
    if counts is None:
        counts = [n] * world
    if all(k == counts[0] for k in counts):
        out = torch.empty((world * n, c), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
        return (out, work) if async_op else out
    if async_op:
        raise ValueError("swnerf.parallel.gather_pixels: async_op needs equal shards")
    m = max(counts)
    pad = torch.zeros((m, c), dtype=local.dtype, device=local.device)
    pad[:n] = local
    out = torch.empty((world * m, c), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * m:r * m + counts[r]] for r in range(world)], 0)


def render_image_sharded(render_range, H, W, group=None):
    """render_range(ray0, n) -> [n, C] pixels of the row-major ray range; every rank returns the
    whole [H, W, C] image."""
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    ranges = [shard_range(H * W, world, r) for r in range(world)]
    lo, hi = ranges[rank]
    local = render_range(lo, hi - lo)
    full = gather_pixels(local, [b - a for a, b in ranges], group)
    return full.reshape(H, W, -1)


FRAME_STREAMS = int(os.environ.get("SWNERF_FRAME_STREAMS", "1"))   # sub-ranges of a shard rendered concurrently (1: off, the default)
_FRAME_SIDE = {}


def frame_renderer(H, W, K, c2w, render_kwargs, frame_time=None, chunk=1 << 30, device=None, streams=None):
    """The per-rank half of the sharded full-image render (BASELINE configs C4 / C5; the reference's render-only
    entry nerf/run.py:557-571, d_nerf/run_dnerf.py:553-566 renders every pose with render(H, W, K, c2w=...)):
    returns render_range(ray0, n) -> [n, 5] = [rgb(3), disp, acc] of the row-major pixel range, doing what render()
    does for those pixels - get_rays on the range (each rank generates its own rays from (K, c2w): no scatter), ray
    batch, coarse pass, resampling, fine pass.  frame_time given -> the D-NeRF render (run_dnerf.py:104-173).

    streams > 1 (default FRAME_STREAMS = 1: off): a range of >= 8192 rays is rendered as that many contiguous sub-ranges on
    side streams.  One wavefront owns one ray and 1024 are resident, so a launch of n rays leaves its last round of workgroups
    partly empty (20 000 rays = 19.53 rounds: 2.3 % of the launch idle, twice per render); with two sub-ranges in flight the
    workgroups of one sub-range's next launch can start on the CUs the other's last round leaves idle.  Measured round 3
    (`profiles/r03/frame_streams.md`): +0.25 % on the C4 shard, nothing on the C5 shard, three streams slower - the overlap
    puts two nets' weight streams (4.8 MB) into the 4 MB L2s at once, which costs what the filled tail gains.  Kept as an
    option; rays are independent, so the pixels are the same bits either way (tests/test_gpu_sharded.py)."""
    from . import render as _r, render_dnerf as _rd
    from .ray import get_rays_range
    n_streams = FRAME_STREAMS if streams is None else int(streams)

    def one(ray0, n):
        if frame_time is None:
            o, d = get_rays_range(H, W, K, c2w, ray0, n, device)
            rgb, disp, acc, _ = _r.render(H, W, K, chunk=chunk, rays=(o, d), **render_kwargs)
        else:
            focal = float(K[0][0]) if not isinstance(K, float) else K
            o, d = get_rays_range(H, W, focal, c2w, ray0, n, device)
            rgb, disp, acc, _ = _rd.render(H, W, focal, chunk=chunk, rays=(o, d), frame_time=frame_time, **render_kwargs)
        return torch.cat([rgb, disp[:, None], acc[:, None]], -1)

    def render_range(ray0, n):
        with torch.no_grad():
            dev = torch.device(device) if device is not None else None
            if n_streams <= 1 or n < 8192 or dev is None or dev.type != "cuda":
                return one(ray0, n)
            main = torch.cuda.current_stream(dev)
            if not any(p_.is_cuda for net in (render_kwargs.get("network_fn"), render_kwargs.get("network_fine")) if net is not None
                       for p_ in net.parameters()):
                return one(ray0, n)
            _r.prepack(render_kwargs.get("network_fn"), render_kwargs.get("network_fine"))
            key = (dev.index, n_streams)
            if key not in _FRAME_SIDE:
                _FRAME_SIDE[key] = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
            side = _FRAME_SIDE[key]
            ev0 = torch.cuda.Event()
            ev0.record(main)
            outs, a = [], 0
            for i, s in enumerate(side):
                cnt = (n - a) if i == n_streams - 1 else ((n // n_streams + 3) // 4 * 4)     # whole workgroups of 4 rays
                s.wait_event(ev0)
                with torch.cuda.stream(s):
                    t = one(ray0 + a, cnt)
                t.record_stream(main)
                outs.append(t)
                a += cnt
            for s in side:
                ev = torch.cuda.Event()
                ev.record(s)
                main.wait_event(ev)
            return torch.cat(outs, 0)
    return render_range


class GradBucket:
    """Data-parallel training without the gather / scatter of allreduce_gradients: the gradients of each module LIVE in one
    flat bucket (`p.grad` are views of it), so the collective is ONE in-place all_reduce per module and nothing is copied
    around it.  The all-reduce of a module is issued - async, on the backend's own stream behind the kernels that produced the
    gradients - the moment autograd has accumulated its last parameter (post-accumulate hooks): for the reference's step
    (nerf/run.py:684-708, loss = mse(rgb) + mse(rgb0)) the fine net's 2.4 MB bucket is reduced over xGMI while the coarse
    pass's backward still runs; wait() blocks only before optimizer.step().

        bucket = GradBucket([coarse, fine])            # once
        bucket.zero(); loss.backward(); bucket.wait(); optimizer.step()

    A parameter that got no gradient in a step (a D-NeRF batch at frame_time == 0 leaves `_time.*` untouched,
    model.py:143-145) contributes its zeros: every rank reduces the same element count by construction.  wait() launches the
    buckets whose hooks never completed.  force: run the collective in a group of one rank too (bench.py --collective always:
    the code path of an N-GPU job on a one-GPU box).  No process group and no force: the bucket is only a gradient arena."""

    def __init__(self, modules, group=None, average=True, force=False):
        self.group, self.average, self.force = group, average, force
        self.items = []                                  # per module: dict(flat, params, seen, work)
        for m in modules:
            if m is None:
                continue
            params = [p for p in m.parameters() if p.requires_grad]
            if not params:
                continue
            offs, n = [], 0
            for p in params:
                offs.append(n)
                n += (p.numel() + 3) // 4 * 4            # every view 16-byte aligned
            flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
            item = {"flat": flat, "params": params, "seen": 0, "work": None, "launched": False}
            for p, o in zip(params, offs):
                p.grad = flat[o:o + p.numel()].view_as(p)
                p.register_post_accumulate_grad_hook(lambda _p, it=item: self._accumulated(it))
            self.items.append(item)

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (self.force or dist.get_world_size(self.group) > 1)

    def _launch(self, it):
        it["launched"] = True
        if self._active():
            it["work"] = dist.all_reduce(it["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _accumulated(self, it):
        it["seen"] += 1
        if it["seen"] == len(it["params"]) and not it["launched"]:
            self._launch(it)

    def zero(self):
        """Start of a step (instead of optimizer.zero_grad(): the gradients must stay views of the bucket)."""
        for it in self.items:
            it["flat"].zero_()
            it["seen"], it["work"], it["launched"] = 0, None, False
            for p in it["params"]:
                if p.grad is None or p.grad.data_ptr() < it["flat"].data_ptr() or p.grad.data_ptr() >= it["flat"].data_ptr() + it["flat"].numel() * it["flat"].element_size():
                    raise RuntimeError("swnerf.parallel.GradBucket: a parameter's .grad no longer lives in the bucket "
                                       "(use bucket.zero() instead of optimizer.zero_grad(set_to_none=True))")

    def wait(self):
        """End of backward: every bucket reduced (and averaged), safe to read `p.grad` / call optimizer.step()."""
        for it in self.items:
            if not it["launched"]:
                self._launch(it)
        for it in self.items:
            if it["work"] is not None:
                it["work"].wait()
                it["work"] = None
                world = dist.get_world_size(self.group)
                if self.average and world > 1:
                    it["flat"].div_(world)

    def nbytes(self):
        return sum(it["flat"].numel() * it["flat"].element_size() for it in self.items)


def allreduce_gradients(modules, group=None, average=True, force=False):
    """Data-parallel training over the ray batch (SURVEY.md section 8e "Training"): ONE all-reduce of every
    parameter gradient, flattened into a single bucket (2 x 595 844 floats = 4.77 MB for coarse + fine) -
    a single RCCL call instead of 48 small ones; xGMI rings are per-link bound, so few large messages.
    The bucket is built from a RANK-INVARIANT list - every parameter with requires_grad, zeros where this
    rank produced no gradient (a D-NeRF rank whose batch sits at frame_time == 0 takes the zero_canonical
    branch, model.py:143-145, and leaves `_time.*` without .grad) - so every rank contributes the same element
    count; a parameter gets a .grad afterwards if ANY rank had one.
    (GradBucket above is the copy-free form for a training loop that can hand it the modules up front.)"""
    if not (dist.is_available() and dist.is_initialized()):
        if force:
            raise RuntimeError("swnerf.parallel.allreduce_gradients(force=True) needs an initialised process group")
        return
    if dist.get_world_size(group) == 1 and not force:        # force: the collective runs in a group of one rank as well
        return
    params = [p for m in modules if m is not None for p in m.parameters() if p.requires_grad]
    if not params:
        return
    dev, dt = params[0].device, params[0].dtype
    n = sum(p.numel() for p in params)
    # ONE bucket: every gradient (zeros for a parameter this rank has none for) followed by one "has a gradient" flag per
    # parameter.  The flags are built on the host and uploaded as one small tensor; the bucket is one torch.cat - no
    # per-parameter launches beyond the views cat reads from.
    has_local = [p.grad is not None for p in params]
    pieces = [(p.grad if h else torch.zeros_like(p)).reshape(-1).to(dt) for p, h in zip(params, has_local)]
    pieces.append(torch.tensor([1.0 if h else 0.0 for h in has_local], dtype=dt).to(dev, non_blocking=True))
    flat = torch.cat(pieces)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    world = dist.get_world_size(group)
    if average:
        flat[:n] /= world
    if all(has_local):
        has = has_local                                          # every rank sums >= 1 there: no read-back needed
    else:
        has = [v > 0 for v in flat[n:].tolist()]                 # the rare case (a D-NeRF rank at t == 0): one small D2H
    grads = torch.split(flat[:n], [p.numel() for p in params])
    for p, g, h in zip(params, grads, has):
        if not h:
            continue
        if p.grad is None:
            p.grad = g.view_as(p).clone()
        else:
            p.grad.copy_(g.view_as(p))
