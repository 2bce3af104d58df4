"""Dense-grid network query for mesh extraction: the GPU side of
nerf/extract_mesh.py (`generate_viewdirs` :7-24, `sample_grid` :27-90) and of the 2-D form of
`network_query_fn` it uses (nerf/load_model.py:56-74).  Marching cubes / trimesh export stay with
the caller (host tools, out of scope).

`swnerf_query_points` evaluates the positional encodings in registers and - for V view
directions shared by every grid point - the 8-layer trunk and the density ONCE per point and
only the view branch V times: 8832 + 640 V MFMAs per 32 points instead of 9472 V."""
import numpy as np
import torch

from . import _lib


def generate_viewdirs(num_views=100):
    """extract_mesh.py:7-24: golden-angle spiral on the unit sphere, float64 [num_views, 3]."""
    k = np.arange(0, num_views, dtype=float) + 0.5
    phi = np.arccos(1 - 2 * k / num_views)
    theta = np.pi * (1 + 5 ** 0.5) * k
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def query_points(net, pts, viewdirs, shared_dirs=None):
    """pts [M,3]; viewdirs [M,3] (one per point -> raw [M,4], what network_query_fn(positions, viewdirs, fn)
    returns) or [V,3] shared (-> [M,4] = [mean_v raw rgb, sigma]).  `shared_dirs` defaults to
    `viewdirs.shape[0] != M`."""
    kind, packed, Lp, Ld, _ = net.packed()
    if kind != _lib.NET_CANON:
        raise NotImplementedError("swnerf.mesh.query_points: static NeRF nets only (the mesh tool is nerf/ only)")
    pts = _lib.dev_f32(pts, "pts", 3).reshape(-1, 3)
    dirs = _lib.dev_f32(viewdirs, "viewdirs", 3).reshape(-1, 3)
    M = pts.shape[0]
    if shared_dirs is None:
        shared_dirs = dirs.shape[0] != M
    out = torch.empty((M, 4), dtype=torch.float32, device=pts.device)
    _lib.check(_lib.lib().swnerf_query_points(_lib.ptr(packed), _lib.ptr(pts), M, _lib.ptr(dirs), dirs.shape[0],
                                              int(bool(shared_dirs)), Lp, Ld, _lib.ptr(out), _lib.stream_of(pts)),
               "query_points")
    return out


def sample_grid(bounds, resolution, net, num_views=100, batch_size=1 << 20, sharded=None, group=None, query=None):
    """extract_mesh.py:27-90 with the network in place of `nerf_function`:
    -> (density_field [R,R,R], color_field [R,R,R,3], (X, Y, Z)), float64 numpy like the reference.
    `color` is the view-average of the RAW rgb and `density` of the raw sigma (batch_query_fn :155-175
    applies no sigmoid / relu).
    Multi-GPU ("shards over 8 GPUs the same way"): with torch.distributed initialised (`sharded`
    defaults to that) every rank queries its contiguous shard of the R^3 points and ONE all-gather of the [n,4] results
    returns the whole field to every rank - points are independent, exactly like rays (swnerf.parallel).
    `query(points [n,3] tensor, dirs [V,3] tensor) -> [n,4]` defaults to the fused HIP query of `net`."""
    import torch.distributed as dist
    from .parallel import gather_pixels
    from .synth import shard_range
    x = np.linspace(bounds[0][0], bounds[0][1], resolution)
    y = np.linspace(bounds[1][0], bounds[1][1], resolution)
    z = np.linspace(bounds[2][0], bounds[2][1], resolution)
    X, Y, Z = np.meshgrid(x, y, z, indexing='ij')
    points = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=-1)
    dev = next(net.parameters()).device if net is not None else torch.device("cpu")
    dirs = torch.tensor(generate_viewdirs(num_views), dtype=torch.float32, device=dev)
    if query is None:
        query = lambda p, d: query_points(net, p, d, shared_dirs=True)
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if sharded is None:
        sharded = on
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if (sharded and on) else (1, 0)
    ranges = [shard_range(len(points), world, r) for r in range(world)]
    lo, hi = ranges[rank]
    outs = []
    with torch.no_grad():
        for s in range(lo, hi, batch_size):
            p = torch.tensor(points[s:min(hi, s + batch_size)], dtype=torch.float32, device=dev)
            outs.append(query(p, dirs))
    local = torch.cat(outs, 0) if outs else torch.empty((0, 4), dtype=torch.float32, device=dev)
    out = gather_pixels(local, [b - a for a, b in ranges], group) if world > 1 else local
    out = out.cpu().numpy().astype(np.float64)
    return (out[:, 3].reshape(resolution, resolution, resolution),
            out[:, :3].reshape(resolution, resolution, resolution, 3), (X, Y, Z))
