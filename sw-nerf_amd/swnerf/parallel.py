"""Ray-sharded data parallelism.  The reference has no distributed
code; rays are independent, so the full-image render is split into contiguous row-major
shards, one process per GPU, and the only exchange is ONE all-gather of the rendered
pixels [rgb(3), disp, acc] (1.6 MB per rank for an 800x800 frame on 8 GPUs) - RCCL over
xGMI when the process group's backend is "nccl", gloo in the CPU tests."""
import torch
import torch.distributed as dist

from .synth import shard_range


def gather_pixels(local, counts=None, group=None):
    """all-gather of per-rank [n_local, C] pixel blocks into [sum n, C] on every rank.
    Equal shards use a single all_gather_into_tensor (one RCCL call); ragged shards pad to the
    largest and trim."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n, c = local.shape
    if counts is None:
        counts = [n] * world
    if all(k == counts[0] for k in counts):
        out = torch.empty((world * n, c), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
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


def allreduce_gradients(modules, group=None, average=True):
    """Data-parallel training over the ray batch (SURVEY.md section 8e "Training"): ONE all-reduce of every
    parameter gradient, flattened into a single bucket (2 x 595 844 floats = 4.77 MB for coarse + fine) -
    a single RCCL call instead of 48 small ones; xGMI rings are per-link bound, so few large messages."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    params = [p for m in modules if m is not None for p in m.parameters() if p.grad is not None]
    if not params:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
