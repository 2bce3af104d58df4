"""CPU oracle for the NeRF volumetric-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, op for op and unfused, what daihangpku/SW-NeRF computes on the
path  get_rays -> coarse sampling -> positional encoding -> 8x256 MLP ->
raw2outputs -> sample_pdf -> fine pass  (SURVEY.md section 8a).  It runs on the
host in float32 with torch CPU tensors (the reference's own arithmetic is PyTorch
ATen CPU kernels) and is used ONLY as a checker:

  * tests/            compare the HIP path against it,
  * __graft_entry__.smoke()  checks one tiny launch against it,
  * bench.py          times it as the `cpu_baseline` leg (kind="port").

Nothing under sw-nerf_amd/ may import this module; the product path has no CPU
fallback and fails loudly when libswnerf_hip.so is missing.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below
against tests/golden/*.npz, which tests/golden/make_golden.py captured by
importing the unmodified reference (/root/reference) in the build container.

Citations are file:line relative to /root/reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- rays


def get_rays(H, W, focal_or_K, c2w):
    """ray.py:10-38.  Pixel centres are integers (no +0.5); camera looks down -z."""
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    xs = torch.linspace(0, W - 1, W)
    ys = torch.linspace(0, H - 1, H)
    px = xs[None, :].expand(H, W)
    py = ys[:, None].expand(H, W)
    if isinstance(focal_or_K, float):
        cam_x = (px - W * 0.5) / focal_or_K
        cam_y = -(py - H * 0.5) / focal_or_K
    else:
        Kmat = focal_or_K
        cam_x = (px - Kmat[0][2]) / Kmat[0][0]
        cam_y = -(py - Kmat[1][2]) / Kmat[1][1]
    cam = torch.stack([cam_x, cam_y, -torch.ones_like(px)], -1)          # [H,W,3]
    # world direction = R @ cam, written as the reference's broadcast-multiply-sum
    rays_d = (cam[..., None, :] * c2w[:3, :3]).sum(-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_np(H, W, focal_or_K, c2w):
    """ray.py:42-72 (numpy twin; float32 grid, dtype follows K/c2w promotion)."""
    gx, gy = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    if isinstance(focal_or_K, float):
        cam = np.stack([(gx - W * 0.5) / focal_or_K, -(gy - H * 0.5) / focal_or_K, -np.ones_like(gx)], -1)
    else:
        Kmat = focal_or_K
        cam = np.stack([(gx - Kmat[0][2]) / Kmat[0][0], -(gy - Kmat[1][2]) / Kmat[1][1], -np.ones_like(gx)], -1)
    rays_d = np.sum(cam[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """ray.py:75-92.  Shift origins to the near plane, then the projective map."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    sx = -1. / (W / (2. * focal))
    sy = -1. / (H / (2. * focal))
    ox, oy, oz = rays_o[..., 0], rays_o[..., 1], rays_o[..., 2]
    # (s*ox)/oz for the origin, s*(dx/dz - ox/oz) for the direction: evaluation
    # order of ray.py:81-87 kept so the float32 roundings are the same
    o = torch.stack([sx * ox / oz, sy * oy / oz, 1. + 2. * near / oz], -1)
    d = torch.stack([sx * (rays_d[..., 0] / rays_d[..., 2] - ox / oz),
                     sy * (rays_d[..., 1] / rays_d[..., 2] - oy / oz),
                     -2. * near / oz], -1)
    return o, d


# ---------------------------------------------------------------------- embedding


def embed(x, multires):
    """embedder.py:12-59 with include_input, log_sampling, [sin, cos]:
    [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], blocks d wide."""
    if multires < 0:
        return x
    parts = [x]
    bands = 2. ** torch.linspace(0., multires - 1, steps=multires) if multires > 0 else []
    for f in bands:
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


def embed_dim(multires, d):
    return d * (1 + 2 * multires)


# ---------------------------------------------------------------------------- MLP


def _lin(sd, name, h):
    return F.linear(h, sd[name + ".weight"], sd[name + ".bias"])


def nerf_mlp(sd, x, input_ch=63, input_ch_views=27, prefix=""):
    """vallina_NeRF.forward (model.py:39-62) == NeRFOriginal.forward (model.py:273-296)
    with D=8, W=256, skips=[4], use_viewdirs=True.  sd: dict name -> float32 tensor."""
    pts, views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    h = pts
    for i in range(8):
        h = F.relu(_lin(sd, f"{prefix}pts_linears.{i}", h))
        if i == 4:
            h = torch.cat([pts, h], -1)
    sigma = _lin(sd, f"{prefix}alpha_linear", h)
    feat = _lin(sd, f"{prefix}feature_linear", h)
    h = F.relu(_lin(sd, f"{prefix}views_linears.0", torch.cat([feat, views], -1)))
    rgb = _lin(sd, f"{prefix}rgb_linear", h)
    return torch.cat([rgb, sigma], -1)


def dnerf_mlp(sd, x, t_emb, multires=10, input_ch=63, input_ch_views=27, zero_canonical=True, dx_value=None):
    """DirectTemporalNeRF.forward (model.py:138-151) + query_time (model.py:128-136).
    t_emb: [M, 21] embedded frame time (all rows the same time).  Returns (out[M,4], dx[M,3]).
    dx_value (tests only): substitute these VALUES for dx before the re-embedding, gradients untouched - the top
    band of gamma() multiplies a 1e-7 difference in dx by 512, so gradient parity is checked at equal dx."""
    pts, views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    cur_time = float(t_emb[0, 0])
    if cur_time == 0. and zero_canonical:
        dx = torch.zeros_like(pts[:, :3])
    else:
        h = torch.cat([pts, t_emb], -1)
        for i in range(8):
            h = F.relu(_lin(sd, f"_time.{i}", h))
            if i == 4:
                h = torch.cat([pts, h], -1)
        dx = _lin(sd, "_time_out", h)
        if dx_value is not None:
            dx = dx + (dx_value - dx).detach()
        pts = embed(pts[:, :3] + dx, multires)
    out = nerf_mlp(sd, torch.cat([pts, views], -1), input_ch, input_ch_views, prefix="_occ.")
    return out, dx


# ---------------------------------------------------------------- volume rendering


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0., white_bkgd=False, noise=None):
    """ray.py:155-198.  `noise` ([N,S], already scaled by raw_noise_std) replaces the
    reference's torch.randn draw (its `pytest=` hook shows that is the only RNG use)."""
    dz = z_vals[..., 1:] - z_vals[..., :-1]
    dz = torch.cat([dz, torch.full_like(dz[..., :1], 1e10)], -1)
    dz = dz * torch.norm(rays_d[..., None, :], dim=-1)
    colour = torch.sigmoid(raw[..., :3])
    density = raw[..., 3]
    if noise is not None:
        density = density + noise
    alpha = 1. - torch.exp(-F.relu(density) * dz)
    trans = torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1)), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = (weights[..., None] * colour).sum(-2)
    depth_map = (weights * z_vals).sum(-1)
    acc_map = weights.sum(-1)
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / weights.sum(-1))
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


def sample_pdf(bins, weights, N_samples, det=False, u=None):
    """ray.py:96-153.  `u` ([N,N_samples]) replaces the torch.rand draw when given."""
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0., 1., steps=N_samples).expand(list(cdf.shape[:-1]) + [N_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [N_samples])
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(hi - 1, min=0)
    hi = torch.clamp(hi, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    bin_lo, bin_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    span = cdf_hi - cdf_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return bin_lo + (u - cdf_lo) / span * (bin_hi - bin_lo)


def coarse_z(near, far, N_samples, lindisp=False, t_rand=None):
    """nerf/run.py:361-383.  near/far: [N,1].  t_rand ([N,S]) = injected stratified jitter."""
    t = torch.linspace(0., 1., steps=N_samples)
    if not lindisp:
        z = near * (1. - t) + far * t
    else:
        z = 1. / (1. / near * (1. - t) + 1. / far * t)
    z = z.expand([near.shape[0], N_samples])
    if t_rand is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


def run_network(sd, pts, viewdirs, multires=10, multires_views=4, netchunk=1024 * 64):
    """nerf/run.py:73-87 with the standard embedders; chunked like `batchify` (:63-70)."""
    flat = pts.reshape(-1, pts.shape[-1])
    e = embed(flat, multires)
    dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
    e = torch.cat([e, embed(dirs, multires_views)], -1)
    out = torch.cat([nerf_mlp(sd, e[i:i + netchunk], embed_dim(multires, 3), embed_dim(multires_views, 3))
                     for i in range(0, e.shape[0], netchunk)], 0)
    return out.reshape(list(pts.shape[:-1]) + [4])


def run_network_dnerf(sd, pts, viewdirs, frame_time, multires=10, multires_views=4,
                      netchunk=1024 * 64, zero_canonical=True):
    """d_nerf/run_dnerf.py:46-83."""
    flat = pts.reshape(-1, 3)
    e = embed(flat, multires)
    B, S, _ = pts.shape
    t_emb = embed(frame_time[:, None].expand(B, S, 1).reshape(-1, 1), multires)
    dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
    e = torch.cat([e, embed(dirs, multires_views)], -1)
    outs, dxs = [], []
    for i in range(0, e.shape[0], netchunk):
        o, dx = dnerf_mlp(sd, e[i:i + netchunk], t_emb[i:i + netchunk], multires,
                          embed_dim(multires, 3), embed_dim(multires_views, 3), zero_canonical)
        outs.append(o)
        dxs.append(dx)
    return torch.cat(outs, 0).reshape(B, S, 4), torch.cat(dxs, 0).reshape(B, S, 3)


def render_rays(ray_batch, sd_coarse, sd_fine, N_samples, N_importance=0, lindisp=False,
                white_bkgd=False, retraw=False, t_rand=None, u=None, noise0=None, noise1=None,
                multires=10, multires_views=4):
    """nerf/run.py:316-422 (static NeRF).  ray_batch [N,11] = o,d,near,far,viewdirs.
    perturb>0 <=> t_rand given (and then u must be given too: det = (perturb==0))."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:]
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z = coarse_z(near, far, N_samples, lindisp, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    raw = run_network(sd_coarse, pts, viewdirs, multires, multires_views)
    rgb, disp, acc, weights, _ = raw2outputs(raw, z, rays_d, 0., white_bkgd, noise0)
    ret = {}
    if N_importance > 0:
        ret.update(rgb0=rgb, disp0=disp, acc0=acc)
        mid = .5 * (z[..., 1:] + z[..., :-1])
        zs = sample_pdf(mid, weights[..., 1:-1], N_importance, det=(t_rand is None), u=u)
        z, _ = torch.sort(torch.cat([z, zs], -1), -1)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
        raw = run_network(sd_fine if sd_fine is not None else sd_coarse, pts, viewdirs, multires, multires_views)
        rgb, disp, acc, weights, _ = raw2outputs(raw, z, rays_d, 0., white_bkgd, noise1)
        ret["z_std"] = torch.std(zs, dim=-1, unbiased=False)
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc)
    ret["z_vals"] = z          # not returned by the static reference; kept for parity checks
    if retraw:
        ret["raw"] = raw
    return ret


def render_rays_dnerf(ray_batch, sd, N_samples, N_importance=0, lindisp=False, white_bkgd=False,
                      retraw=False, t_rand=None, u=None, multires=10, multires_views=4,
                      zero_canonical=True):
    """d_nerf/run_dnerf.py:354-480, single model (use_two_models_for_fine=False), z_vals=None.
    ray_batch [N,12] = o,d,near,far,time,viewdirs."""
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:]
    near, far, ft = ray_batch[:, 6:7], ray_batch[:, 7:8], ray_batch[:, 8:9]
    z = coarse_z(near, far, N_samples, lindisp, t_rand)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    ret = {}
    if N_importance > 0:
        raw, _ = run_network_dnerf(sd, pts, viewdirs, ft, multires, multires_views, zero_canonical=zero_canonical)
        _, _, _, weights, _ = raw2outputs(raw, z, rays_d, 0., white_bkgd)
        mid = .5 * (z[..., 1:] + z[..., :-1])
        zs = sample_pdf(mid, weights[..., 1:-1], N_importance, det=(t_rand is None), u=u)
        z, _ = torch.sort(torch.cat([z, zs], -1), -1)
        ret["z_std"] = torch.std(zs, dim=-1, unbiased=False)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    raw, dx = run_network_dnerf(sd, pts, viewdirs, ft, multires, multires_views, zero_canonical=zero_canonical)
    rgb, disp, acc, _, _ = raw2outputs(raw, z, rays_d, 0., white_bkgd)
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc, z_vals=z, position_delta=dx)
    if retraw:
        ret["raw"] = raw
    return ret


def query_points(sd, pts, viewdirs, multires=10, multires_views=4):
    """network_query_fn on bare points, nerf/load_model.py:56-74 (2-D inputs: pts [M,3], viewdirs [M,3])."""
    e = torch.cat([embed(pts, multires), embed(viewdirs, multires_views)], -1)
    return nerf_mlp(sd, e, embed_dim(multires, 3), embed_dim(multires_views, 3))


def sample_grid_points(sd, points, viewdirs):
    """The inner double loop of nerf/extract_mesh.py:61-79 for points [M,3] (float32) and viewdirs [V,3]:
    per view one network query of all points; colours/densities averaged over the views in float64."""
    cols = np.zeros((points.shape[0], viewdirs.shape[0], 3))
    dens = np.zeros((points.shape[0], viewdirs.shape[0]))
    for v in range(viewdirs.shape[0]):
        out = query_points(sd, points, viewdirs[v][None].expand(points.shape[0], 3)).numpy()
        cols[:, v] = out[:, :3]
        dens[:, v] = out[:, 3]
    return dens.mean(1), cols.mean(1)


def make_ray_batch(rays_o, rays_d, near, far, frame_time=None, ndc=False, H=None, W=None, focal=None):
    """The packing done inside render(): nerf/run.py:137-158, d_nerf/run_dnerf.py:137-160.
    viewdirs are normalised BEFORE the NDC warp."""
    viewdirs = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    viewdirs = viewdirs.reshape(-1, 3).float()
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)
    rays_o = rays_o.reshape(-1, 3).float()
    rays_d = rays_d.reshape(-1, 3).float()
    cols = [rays_o, rays_d, near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])]
    if frame_time is not None:
        cols.append(frame_time * torch.ones_like(rays_d[..., :1]))
    cols.append(viewdirs)
    return torch.cat(cols, -1)


def to_torch_sd(sd_np):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


# ---------------------------------------------------------------- any-shape nets (test infrastructure, like the rest)
def generic_mlp(sd, x, D, skips, input_ch, input_ch_views, use_viewdirs, prefix=""):
    """vallina_NeRF.forward / NeRFOriginal.forward (model.py:39-62, 273-296) for ANY D, W, skips, use_viewdirs
    (use_viewdirs=False: outputs = output_linear(h), model.py:59-60)."""
    pts, views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    h = pts
    for i in range(D):
        h = F.relu(_lin(sd, f"{prefix}pts_linears.{i}", h))
        if i in skips:
            h = torch.cat([pts, h], -1)
    if not use_viewdirs:
        return _lin(sd, f"{prefix}output_linear", h)
    sigma = _lin(sd, f"{prefix}alpha_linear", h)
    feat = _lin(sd, f"{prefix}feature_linear", h)
    h = F.relu(_lin(sd, f"{prefix}views_linears.0", torch.cat([feat, views], -1)))
    return torch.cat([_lin(sd, f"{prefix}rgb_linear", h), sigma], -1)


def generic_dnerf_mlp(sd, x, t_emb, D, skips, input_ch, input_ch_views, use_viewdirs, multires=10, zero_canonical=True):
    """DirectTemporalNeRF.forward (model.py:128-151) for any D, W, skips."""
    pts, views = torch.split(x, [input_ch, input_ch_views], dim=-1)
    if float(t_emb[0, 0]) == 0. and zero_canonical:
        dx = torch.zeros_like(pts[:, :3])
    else:
        h = torch.cat([pts, t_emb], -1)
        for i in range(D):
            h = F.relu(_lin(sd, f"_time.{i}", h))
            if i in skips:
                h = torch.cat([pts, h], -1)
        dx = _lin(sd, "_time_out", h)
        pts = embed(pts[:, :3] + dx, multires)
    return generic_mlp(sd, torch.cat([pts, views], -1), D, skips, input_ch, input_ch_views, use_viewdirs, prefix="_occ."), dx


def render_rays_generic(ray_batch, net_fn, N_samples, N_importance=0, white_bkgd=False, multires=10, multires_views=4, retraw=False):
    """render_rays (nerf/run.py:316-422) with ONE net given as a callable embedded-rows -> raw (any output_ch >= 4), with or
    without view directions (ray_batch 11 or 8 columns), perturb = 0."""
    N = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]

    def query(pts):
        e = embed(pts.reshape(-1, 3), multires)
        if viewdirs is not None:
            e = torch.cat([e, embed(viewdirs[:, None].expand(pts.shape).reshape(-1, 3), multires_views)], -1)
        raw = net_fn(e)
        return raw.reshape(N, -1, raw.shape[-1])

    z = coarse_z(near, far, N_samples)
    raw = query(rays_o[:, None] + rays_d[:, None] * z[..., None])
    rgb, disp, acc, w, _ = raw2outputs(raw[..., :4], z, rays_d, 0., white_bkgd)
    ret = {}
    if N_importance > 0:
        ret.update(rgb0=rgb, disp0=disp, acc0=acc)
        zs = sample_pdf(.5 * (z[:, 1:] + z[:, :-1]), w[:, 1:-1], N_importance, det=True)
        z, _ = torch.sort(torch.cat([z, zs], -1), -1)
        raw = query(rays_o[:, None] + rays_d[:, None] * z[..., None])
        rgb, disp, acc, w, _ = raw2outputs(raw[..., :4], z, rays_d, 0., white_bkgd)
        ret["z_std"] = torch.std(zs, dim=-1, unbiased=False)
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc)
    if retraw:
        ret["raw"] = raw
    return ret


def render_rays_two_nets_generic(ray_batch, coarse_fn, fine_fn, N_samples, N_importance, white_bkgd=False, multires=10, multires_views=4):
    """render_rays (nerf/run.py:316-422) with separate coarse and fine nets given as callables embedded-rows -> raw (any
    output_ch >= 4), with or without view directions (ray_batch 11 or 8 columns), perturb = 0, N_importance > 0."""
    N = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]

    def query(net_fn, pts):
        e = embed(pts.reshape(-1, 3), multires)
        if viewdirs is not None:
            e = torch.cat([e, embed(viewdirs[:, None].expand(pts.shape).reshape(-1, 3), multires_views)], -1)
        raw = net_fn(e)
        return raw.reshape(N, -1, raw.shape[-1])

    z = coarse_z(near, far, N_samples)
    raw = query(coarse_fn, rays_o[:, None] + rays_d[:, None] * z[..., None])
    rgb0, disp0, acc0, w, _ = raw2outputs(raw[..., :4], z, rays_d, 0., white_bkgd)
    zs = sample_pdf(.5 * (z[:, 1:] + z[:, :-1]), w[:, 1:-1], N_importance, det=True)
    z, _ = torch.sort(torch.cat([z, zs], -1), -1)
    raw = query(fine_fn, rays_o[:, None] + rays_d[:, None] * z[..., None])
    rgb, disp, acc, w, _ = raw2outputs(raw[..., :4], z, rays_d, 0., white_bkgd)
    return dict(rgb_map=rgb, disp_map=disp, acc_map=acc, rgb0=rgb0, disp0=disp0, acc0=acc0, z_std=torch.std(zs, dim=-1, unbiased=False), z_vals=z)
