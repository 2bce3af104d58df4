"""Counterparts of the render functions of the reference's D-NeRF runner
(d_nerf/run_dnerf.py:24-235, 354-480): batchify, run_network, batchify_rays, render,
render_rays with the frame_time plumbing.  Same fused dispatch as swnerf.render."""
import os

import numpy as np
import torch

from . import _lib
from .embedder import to8b
from .png import write_png
from .ray import get_rays, sample_pdf, raw2outputs
from .render import fused_plan, render_pass, pack_ray_batch, _rng_inputs, _coarse_z
from .model import DirectTemporalNeRF

DEBUG = False


def batchify(fn, chunk):
    """d_nerf/run_dnerf.py:24-43."""
    if chunk is None:
        return fn

    def ret(inputs_pos, inputs_time):
        outs, dxs = [], []
        for i in range(0, inputs_pos.shape[0], chunk):
            out, dx = fn(inputs_pos[i:i + chunk], [inputs_time[0][i:i + chunk], inputs_time[1][i:i + chunk]])
            outs.append(out)
            dxs.append(dx)
        return torch.cat(outs, 0), torch.cat(dxs, 0)
    return ret


def run_network(inputs, viewdirs, frame_time, fn, embed_fn, embeddirs_fn, embedtime_fn, netchunk=1024 * 64,
                embd_time_discr=True):
    """d_nerf/run_dnerf.py:46-83.  (The single-time assertion is made once, inside the module.)"""
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if not embd_time_discr:
        raise NotImplementedError
    B, N, _ = inputs.shape
    embedded_time = embedtime_fn(torch.reshape(frame_time[:, None].expand([B, N, 1]), [-1, 1]))
    if viewdirs is not None:
        input_dirs_flat = torch.reshape(viewdirs[:, None].expand(inputs.shape), [-1, viewdirs.shape[-1]])
        embedded = torch.cat([embedded, embeddirs_fn(input_dirs_flat)], -1)
    if isinstance(fn, DirectTemporalNeRF) and fn._wants_grad():
        # training: `netchunk` only bounds the reference's activation memory; one call lets the weight-gradient GEMMs run
        # over all rows at once (same reasoning as swnerf.render.run_network).  Results are identical either way.
        outputs_flat, dx_flat = fn(embedded, [embedded_time, embedded_time])
    else:
        outputs_flat, dx_flat = batchify(fn, netchunk)(embedded, [embedded_time, embedded_time])
    outputs = torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])
    return outputs, torch.reshape(dx_flat, list(inputs.shape[:-1]) + [dx_flat.shape[-1]])


# frame_time known on the host (render() was handed a Python float, the normal case: d_nerf/run_dnerf.py:208,667 pass
# one time per frame): keyed by the ray batch's storage, valid only while render() runs, so that render_rays - whose
# signature is the reference's and carries the time only as column 8 - needs no device->host sync per chunk.
_TIME_HINT = {}


def _single_time(ray_batch):
    hint = _TIME_HINT.get(ray_batch.untyped_storage().data_ptr())
    if hint is not None:
        return hint
    lo, hi = torch.aminmax(ray_batch[:, 8])
    lo, hi = float(lo), float(hi)
    assert lo == hi, "Only accepts all points from same time"      # run_dnerf.py:53
    return lo


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                z_vals=None, use_two_models_for_fine=False):
    """d_nerf/run_dnerf.py:354-480."""
    plan = None
    if ray_batch.shape[-1] == 12:
        plan = fused_plan(network_query_fn, [network_fn, network_fine], need_time=True)
    if plan is None:
        return _render_rays_unfused(ray_batch, network_fn, network_query_fn, N_samples, retraw, lindisp, perturb,
                                    N_importance, network_fine, white_bkgd, raw_noise_std, pytest, z_vals,
                                    use_two_models_for_fine)
    N = ray_batch.shape[0]
    t0 = _single_time(ray_batch)
    deform = lambda net: isinstance(net, DirectTemporalNeRF) and not (t0 == 0. and net.zero_canonical)
    t_rand, u, noise = _rng_inputs(N, N_samples, N_importance, perturb, raw_noise_std, pytest, ray_batch.device)
    run_fn = network_fn if network_fine is None else network_fine
    full = ["rgb_map", "disp_map", "acc_map", "dx"]
    z_std = None
    p0 = None
    if z_vals is None:
        if N_importance <= 0:
            # (the reference evaluates the net twice here with identical results, run_dnerf.py:434-436,455-458)
            p1 = render_pass(ray_batch, run_fn, N_samples, lindisp=lindisp, t_rand=t_rand, noise=noise(N_samples),
                             white_bkgd=white_bkgd, want=full + ["z_out"] + (["raw"] if retraw else []),
                             run_deform=deform(run_fn))
            z_final = p1["z_out"]
        else:
            want0 = ["rgb_map", "disp_map", "acc_map", "dx"] if use_two_models_for_fine else []
            p0 = render_pass(ray_batch, network_fn, N_samples, lindisp=lindisp, t_rand=t_rand, noise=noise(N_samples),
                             white_bkgd=white_bkgd, want=want0, n_importance=N_importance, u=u,
                             run_deform=deform(network_fn))
            z_final, z_std = p0["z_fine"], p0["z_std"]
            p1 = None
    else:
        z_final = _lib.dev_f32(z_vals, "z_vals")
        p1 = None
    if p1 is None:
        S1 = z_final.shape[-1]
        p1 = render_pass(ray_batch, run_fn, S1, z_vals=z_final, noise=noise(S1), white_bkgd=white_bkgd,
                         want=full + (["raw"] if retraw else []), run_deform=deform(run_fn))
    ret = {'rgb_map': p1["rgb_map"], 'disp_map': p1["disp_map"], 'acc_map': p1["acc_map"], 'z_vals': z_final,
           'position_delta': p1["dx"]}
    if retraw:
        ret['raw'] = p1["raw"]
    if N_importance > 0:
        if p0 is not None and use_two_models_for_fine:
            ret['rgb0'], ret['disp0'], ret['acc0'] = p0["rgb_map"], p0["disp_map"], p0["acc_map"]
            ret['position_delta_0'] = p0["dx"]
        if z_std is not None:
            ret['z_std'] = z_std
    return ret


def _render_rays_unfused(ray_batch, network_fn, network_query_fn, N_samples, retraw, lindisp, perturb, N_importance,
                         network_fine, white_bkgd, raw_noise_std, pytest, z_vals, use_two_models_for_fine):
    """The reference's op sequence (run_dnerf.py:397-474) on the individual HIP ops."""
    N_rays = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 9 else None
    bounds = torch.reshape(ray_batch[..., 6:9], [-1, 1, 3])
    near, far, frame_time = bounds[..., 0], bounds[..., 1], bounds[..., 2]
    z_samples = None
    rgb_map_0 = disp_map_0 = acc_map_0 = position_delta_0 = None
    fused_coarse = None
    if z_vals is None and N_importance > 0 and not use_two_models_for_fine and ray_batch.shape[-1] == 12:
        # The coarse pass of the one-model configuration only feeds the resampling and runs under no_grad in the
        # reference (run_dnerf.py:417-421): even in a training step it can be the fused HIP pass.
        with torch.no_grad():
            if fused_plan(network_query_fn, [network_fn], need_time=True) is not None:
                t0 = _single_time(ray_batch)
                deform = isinstance(network_fn, DirectTemporalNeRF) and not (t0 == 0. and network_fn.zero_canonical)
                t_rand, u, noise = _rng_inputs(N_rays, N_samples, N_importance, perturb, raw_noise_std, pytest, ray_batch.device)
                fused_coarse = render_pass(ray_batch.detach(), network_fn, N_samples, lindisp=lindisp, t_rand=t_rand,
                                           noise=noise(N_samples), white_bkgd=white_bkgd, want=[], n_importance=N_importance,
                                           u=u, run_deform=deform)
    z_std = None
    if fused_coarse is not None:
        z_vals, z_std = fused_coarse["z_fine"], fused_coarse["z_std"]
    elif z_vals is None:
        z_vals = _coarse_z(near, far, N_rays, N_samples, lindisp, perturb, pytest, ray_batch)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        if N_importance > 0:
            if use_two_models_for_fine:
                raw, position_delta_0 = network_query_fn(pts, viewdirs, frame_time, network_fn)
                rgb_map_0, disp_map_0, acc_map_0, weights, _ = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
            else:
                with torch.no_grad():
                    raw, _ = network_query_fn(pts, viewdirs, frame_time, network_fn)
                    _, _, _, weights, _ = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
            z_vals_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
            z_samples = sample_pdf(z_vals_mid, weights[..., 1:-1], N_importance, det=(perturb == 0.), pytest=pytest).detach()
            z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    run_fn = network_fn if network_fine is None else network_fine
    raw, position_delta = network_query_fn(pts, viewdirs, frame_time, run_fn)
    rgb_map, disp_map, acc_map, weights, _ = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map, 'z_vals': z_vals, 'position_delta': position_delta}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        for k, v in (('rgb0', rgb_map_0), ('disp0', disp_map_0), ('acc0', acc_map_0), ('position_delta_0', position_delta_0)):
            if v is not None:
                ret[k] = v
        if z_samples is not None:
            ret['z_std'] = torch.std(z_samples, dim=-1, unbiased=False)
        elif z_std is not None:
            ret['z_std'] = z_std
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """d_nerf/run_dnerf.py:86-101."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., frame_time=None,
           use_viewdirs=False, c2w_staticcam=None, **kwargs):
    """d_nerf/run_dnerf.py:104-173 -> [rgb_map, disp_map, acc_map, extras]."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, float(focal), c2w)
    else:
        rays_o, rays_d = rays
    viewsrc = rays_d
    if c2w_staticcam is not None:
        rays_o, rays_d = get_rays(H, W, float(focal), c2w_staticcam)
    sh = rays_d.shape
    ft = float(frame_time) if not isinstance(frame_time, torch.Tensor) or frame_time.numel() == 1 else frame_time
    rb = pack_ray_batch(rays_o, rays_d, near, far, frame_time=ft, ndc=ndc, H=H, W=W, focal=focal)
    if c2w_staticcam is not None:
        rb[:, -3:] = pack_ray_batch(rays_o, viewsrc, near, far)[:, -3:]
    if not use_viewdirs:
        rb = rb[:, :9].contiguous()          # rays = cat[o, d, near, far, frame_time] (run_dnerf.py:153-159)
    key = rb.untyped_storage().data_ptr()
    if isinstance(ft, float):
        _TIME_HINT[key] = ft
    try:
        all_ret = batchify_rays(rb, chunk, **kwargs)
    finally:
        _TIME_HINT.pop(key, None)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]


def render_path(render_poses, render_times, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                save_also_gt=False, i_offset=0):
    """d_nerf/run_dnerf.py:175-235: one frame per (pose, time); with `savedir` the frames go to
    savedir/estim/'{:03d}.png' (and the ground truth to savedir/gt/ when save_also_gt)."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    if savedir is not None:
        save_dir_estim, save_dir_gt = os.path.join(savedir, "estim"), os.path.join(savedir, "gt")
        os.makedirs(save_dir_estim, exist_ok=True)
        if save_also_gt:
            os.makedirs(save_dir_gt, exist_ok=True)
    rgbs, disps = [], []
    for i, (c2w, frame_time) in enumerate(zip(render_poses, render_times)):
        rgb, disp, acc, _ = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], frame_time=frame_time, **render_kwargs)
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        if savedir is not None:
            write_png(os.path.join(save_dir_estim, '{:03d}.png'.format(i + i_offset)), to8b(rgbs[-1]))
            if save_also_gt:
                gt = gt_imgs[i]
                gt = gt.cpu().numpy() if isinstance(gt, torch.Tensor) else np.asarray(gt)
                write_png(os.path.join(save_dir_gt, '{:03d}.png'.format(i + i_offset)), to8b(gt))
    return np.stack(rgbs, 0), np.stack(disps, 0)
