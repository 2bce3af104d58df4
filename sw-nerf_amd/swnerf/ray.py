"""Drop-in for the reference's ray.py (ray.py:1-198): get_rays, get_rays_np, ndc_rays,
sample_pdf, raw2outputs with the reference's signatures, executed by HIP kernels through
the C ABI (include/swnerf.h).  Tensors must live on the GPU; there is no CPU path."""
import ctypes
import torch
import torch.nn as nn            # noqa: F401  (re-exported like the reference module does)
import torch.nn.functional as F  # noqa: F401
import numpy as np

from . import _lib
from .embedder import img2mse, mse2psnr, to8b  # noqa: F401


def _device_of(*ts):
    for t in ts:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("swnerf: no GPU visible - the render path runs only on MI355X (HIP); no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _intrinsics(H, W, focal_or_K):
    if isinstance(focal_or_K, float):
        return float(focal_or_K), float(focal_or_K), W * 0.5, H * 0.5, 1
    K = focal_or_K
    return float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), 0


def _c2w_host(c2w):
    m = c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else np.asarray(c2w)
    m = np.ascontiguousarray(m[:3, :4], dtype=np.float32)
    return m


def get_rays_range(H, W, focal_or_K, c2w, ray0, n, device=None, want_origins=True):
    """Rays of the pixels [ray0, ray0+n) in row-major order -> ([n,3], [n,3]).  This is the
    per-rank entry for the sharded full-image render: no scatter needed."""
    dev = device or _device_of(c2w)
    fx, fy, cx, cy, fb = _intrinsics(H, W, focal_or_K)
    m = _c2w_host(c2w)
    rays_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
    rays_o = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_origins else None
    _lib.check(_lib.lib().swnerf_get_rays(int(H), int(W), fx, fy, cx, cy, fb,
                                          m.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(ray0), int(n),
                                          _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.stream_of(rays_d)), "get_rays")
    return rays_o, rays_d


def get_rays(H, W, focal_or_K, c2w):
    """ray.py:10-38.  rays_o is a stride-0 expand of the camera centre, like the reference."""
    dev = _device_of(c2w)
    _, rays_d = get_rays_range(H, W, focal_or_K, c2w, 0, H * W, dev, want_origins=False)
    rays_d = rays_d.reshape(H, W, 3)
    centre = torch.as_tensor(_c2w_host(c2w)[:3, -1].copy(), device=dev)
    return centre.expand(rays_d.shape), rays_d


def get_rays_np(H, W, focal_or_K, c2w):
    """ray.py:42-72 - the host/numpy twin used to pre-build training rays (nerf/run.py:604)."""
    gx, gy = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    if isinstance(focal_or_K, float):
        cam = np.stack([(gx - W * 0.5) / focal_or_K, -(gy - H * 0.5) / focal_or_K, -np.ones_like(gx)], -1)
    else:
        K = focal_or_K
        cam = np.stack([(gx - K[0][2]) / K[0][0], -(gy - K[1][2]) / K[1][1], -np.ones_like(gx)], -1)
    rays_d = np.sum(cam[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """ray.py:75-92."""
    rays_d = _lib.dev_f32(rays_d, "rays_d", 3)
    rays_o = _lib.dev_f32(rays_o.expand(rays_d.shape) if rays_o.shape != rays_d.shape else rays_o, "rays_o", 3)
    o = torch.empty_like(rays_d)
    d = torch.empty_like(rays_d)
    n = rays_d.numel() // 3
    _lib.check(_lib.lib().swnerf_ndc_rays(int(H), int(W), float(focal), float(near), _lib.ptr(rays_o), _lib.ptr(rays_d),
                                          n, _lib.ptr(o), _lib.ptr(d), _lib.stream_of(rays_d)), "ndc_rays")
    return o, d


SAMPLE_PDF_MAX_BINS = 1024          # csrc/misc_kernels.hip SP_MAX_BINS: the kernel keeps a ray's cdf in LDS


def _sample_pdf_wide(bins, weights, N_samples, u):
    """More bins than the kernel's LDS slice holds (N_samples > 1025 coarse samples; the reference takes any count,
    ray.py:96-153), or the degenerate single bin: the same inverse-CDF arithmetic as device tensor ops."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if u is None:
        u = torch.linspace(0., 1., steps=N_samples, device=bins.device).expand(bins.shape[0], N_samples)
    idx = torch.searchsorted(cdf, u.contiguous(), right=True)
    lo, hi = (idx - 1).clamp_min(0), idx.clamp_max(cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    span = c_hi - c_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)
    return b_lo + (u - c_lo) / span * (b_hi - b_lo)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    """ray.py:96-153.  `u` (extra, optional) injects the uniforms instead of torch.rand."""
    bins = _lib.dev_f32(bins, "bins")
    weights = _lib.dev_f32(weights, "weights")
    if bins.dim() != 2 or weights.shape != (bins.shape[0], bins.shape[1] - 1):
        raise ValueError(f"swnerf.sample_pdf: bins [N,M] / weights [N,M-1] expected, got {tuple(bins.shape)} / {tuple(weights.shape)}")
    N, nb = bins.shape
    if u is None and not det:
        u = torch.rand((N, N_samples), device=bins.device)
    if pytest:                      # the reference's determinism hook (ray.py:124-132)
        np.random.seed(0)
        u = None if det else torch.Tensor(np.random.rand(N, N_samples)).to(bins.device)
    if u is not None:
        u = _lib.dev_f32(u, "u", N_samples)
    if nb < 2 or nb > SAMPLE_PDF_MAX_BINS:
        return _sample_pdf_wide(bins, weights, int(N_samples), u)
    samples = torch.empty((N, N_samples), dtype=torch.float32, device=bins.device)
    _lib.check(_lib.lib().swnerf_sample_pdf(_lib.ptr(bins), _lib.ptr(weights), N, nb, int(N_samples), _lib.ptr(u),
                                            _lib.ptr(samples), None, 0, None, None, _lib.stream_of(bins)), "sample_pdf")
    return samples


class _Raw2Outputs(torch.autograd.Function):
    """raw2outputs with the hand-written backward kernel (gradient w.r.t. raw only: z_vals / rays_d are
    detached inputs of the render path, nerf/run.py:398)."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, noise, white_bkgd):
        N, S = z_vals.shape
        dev = raw.device
        rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
        disp = torch.empty((N,), dtype=torch.float32, device=dev)
        acc = torch.empty((N,), dtype=torch.float32, device=dev)
        depth = torch.empty((N,), dtype=torch.float32, device=dev)
        w = torch.empty((N, S), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().swnerf_raw2outputs(_lib.ptr(raw), _lib.ptr(z_vals), _lib.ptr(rays_d), _lib.ptr(noise), N, S,
                                                 int(bool(white_bkgd)), _lib.ptr(rgb), _lib.ptr(disp), _lib.ptr(acc),
                                                 _lib.ptr(w), _lib.ptr(depth), _lib.stream_of(raw)), "raw2outputs")
        ctx.save_for_backward(raw, z_vals, rays_d, noise if noise is not None else torch.empty(0, device=dev))
        ctx.white = bool(white_bkgd)
        ctx.has_noise = noise is not None
        ctx.set_materialize_grads(False)       # an output the loss does not use arrives as None (no zero fill, no read of zeros in the kernel)
        return rgb, disp, acc, w, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
        raw, z_vals, rays_d, noise = ctx.saved_tensors
        N, S = z_vals.shape
        d_raw = torch.empty_like(raw)
        c = lambda g: None if g is None else g.contiguous().float()
        g_rgb, g_disp, g_acc, g_w, g_depth = c(g_rgb), c(g_disp), c(g_acc), c(g_w), c(g_depth)
        _lib.check(_lib.lib().swnerf_raw2outputs_backward(
            _lib.ptr(raw), _lib.ptr(z_vals), _lib.ptr(rays_d), _lib.ptr(noise) if ctx.has_noise else None, N, S, int(ctx.white),
            _lib.ptr(g_rgb), _lib.ptr(g_disp), _lib.ptr(g_acc), _lib.ptr(g_depth), _lib.ptr(g_w), _lib.ptr(d_raw),
            _lib.stream_of(raw)), "raw2outputs_backward")
        return d_raw, None, None, None, None


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, noise=None):
    """ray.py:155-198 -> (rgb_map, disp_map, acc_map, weights, depth_map).
    `noise` (extra, optional) injects the density noise instead of torch.randn * raw_noise_std."""
    if isinstance(raw, torch.Tensor) and raw.shape[-1] > 4:
        raw = raw[..., :4]              # output_ch = 5 (use_viewdirs=False with N_importance > 0, nerf/run.py:231): channels 0..3 are read
    raw = _lib.dev_f32(raw, "raw", 4)
    z_vals = _lib.dev_f32(z_vals, "z_vals")
    rays_d = _lib.dev_f32(rays_d, "rays_d", 3)
    N, S = z_vals.shape
    if raw.shape != (N, S, 4) or rays_d.shape != (N, 3):
        raise ValueError(f"swnerf.raw2outputs: raw [N,S,4], z_vals [N,S], rays_d [N,3] expected, got "
                         f"{tuple(raw.shape)}, {tuple(z_vals.shape)}, {tuple(rays_d.shape)}")
    if noise is None and raw_noise_std > 0.:
        noise = torch.randn((N, S), device=raw.device) * raw_noise_std
        if pytest:                  # ray.py:181-184
            np.random.seed(0)
            noise = torch.Tensor(np.random.rand(N, S) * raw_noise_std).to(raw.device)
    if noise is not None:
        noise = _lib.dev_f32(noise, "noise", S)
    if raw.requires_grad and torch.is_grad_enabled():
        if N == 0:
            raise RuntimeError("swnerf.raw2outputs: empty batch with requires_grad")
        return _Raw2Outputs.apply(raw, z_vals.detach(), rays_d.detach(), noise, white_bkgd)
    dev = raw.device
    rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((N,), dtype=torch.float32, device=dev)
    acc = torch.empty((N,), dtype=torch.float32, device=dev)
    depth = torch.empty((N,), dtype=torch.float32, device=dev)
    w = torch.empty((N, S), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().swnerf_raw2outputs(_lib.ptr(raw), _lib.ptr(z_vals), _lib.ptr(rays_d), _lib.ptr(noise), N, S,
                                             int(bool(white_bkgd)), _lib.ptr(rgb), _lib.ptr(disp), _lib.ptr(acc),
                                             _lib.ptr(w), _lib.ptr(depth), _lib.stream_of(raw)), "raw2outputs")
    return rgb, disp, acc, w, depth
