"""Counterparts of the render functions of the reference's static-NeRF runner
(nerf/run.py:63-219, 316-422): batchify, run_network, batchify_rays, render, render_path,
render_rays - same names, arguments, dict keys and shapes.

render_rays dispatches to the FUSED HIP pass (csrc/render_kernels.hip: one wavefront per
ray, sampling -> encoding -> MLP -> compositing -> resampling without touching HBM) when
`network_fn` is a swnerf module and the `network_query_fn` closure carries the standard
get_embedder encoders; otherwise it runs the reference's op sequence on the individual HIP
ops (embed / mlp_forward / raw2outputs / sample_pdf)."""
import inspect
import os

import numpy as np
import torch

from . import _lib
from .ray import get_rays, sample_pdf, raw2outputs
from .embedder import EmbedFn, to8b
from .png import write_png
from .model import vallina_NeRF, NeRFOriginal, DirectTemporalNeRF

DEBUG = False
PASS_HOOK = None     # bench.py: callable(phase, n_rays, n_samples) with phase in {"begin", "end"} around each launch
Z_TAP = None         # tools/soak_r04.py: a dict that receives the fine depths a render_rays call drew ("fused" / "unfused") - the static
#                      reference does not return them, and a soak must attribute a colour difference to a moved depth before it accepts it


def batchify(fn, chunk):
    """nerf/run.py:63-70."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """nerf/run.py:73-87; also accepts the 2-D form of nerf/load_model.py:56-74 (inputs [N,3],
    viewdirs [N,3] -> [N,1,4]), which goes straight to the fused point query when the encoders and
    the network are this build's."""
    if len(inputs.shape) == 2:
        if (viewdirs is not None and isinstance(embed_fn, EmbedFn) and isinstance(embeddirs_fn, EmbedFn)
                and isinstance(fn, (vallina_NeRF,)) and fn.input_ch == embed_fn.out_dim and fn.input_ch_views == embeddirs_fn.out_dim
                and inputs.is_cuda):
            from .mesh import query_points
            return query_points(fn, inputs, viewdirs, shared_dirs=False)[:, None]
        inputs = inputs[:, None]
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs.shape)
        input_dirs_flat = torch.reshape(input_dirs, [-1, input_dirs.shape[-1]])
        embedded = torch.cat([embedded, embeddirs_fn(input_dirs_flat)], -1)
    if isinstance(fn, (vallina_NeRF,)) and fn._wants_grad():
        # training: `netchunk` only bounds the reference's activation memory; one call lets
        # the weight-gradient GEMMs run over all rows at once.  Results are identical either way.
        outputs_flat = fn(embedded)
    else:
        outputs_flat = batchify(fn, netchunk)(embedded)
    return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


# ------------------------------------------------------------------------------ fused dispatch
def closure_embedders(network_query_fn):
    """The encoders inside the lambda that create_nerf builds (nerf/run.py:248-251,
    d_nerf/run_dnerf.py:281-286), or an explicit `.swnerf_embedders` dict on the callable."""
    tagged = getattr(network_query_fn, "swnerf_embedders", None)
    if tagged is not None:
        return dict(tagged)
    try:
        cv = inspect.getclosurevars(network_query_fn)
    except TypeError:
        return {}
    # create_nerf's lambda closes over locals; a script that builds the same lambda at module level refers to globals
    names = {**cv.globals, **cv.nonlocals}
    return {k: names[k] for k in ("embed_fn", "embeddirs_fn", "embedtime_fn") if k in names}


def wants_grad(nets):
    return torch.is_grad_enabled() and any(p.requires_grad for net in nets if isinstance(net, torch.nn.Module) for p in net.parameters())


def nets_without_views(nets):
    """True when every net given is the 8x256 net WITHOUT view directions the fused pass has a variant for."""
    real = [net for net in nets if net is not None]
    return bool(real) and all(isinstance(net, vallina_NeRF) and net._noview_params() is not None for net in real)


def fused_plan(network_query_fn, nets, need_time=False, allow_train=False):
    """Returns (L_pos, L_dir, L_time) when every net is a swnerf module on the GPU and the
    closure's encoders are the standard ones matching the nets' input sizes; else None.
    Under autograd only callers that have a backward for the fused pass (allow_train: the static render_rays) get a
    plan; everything else takes the differentiable op path."""
    if wants_grad(nets) and not allow_train:
        return None
    emb = closure_embedders(network_query_fn)
    ef, edf, etf = emb.get("embed_fn"), emb.get("embeddirs_fn"), emb.get("embedtime_fn")
    if not (isinstance(ef, EmbedFn) and ef.input_dims == 3):
        return None
    real = [net for net in nets if net is not None]
    if nets_without_views(nets):
        # use_viewdirs=False (the reference's argparse default): no direction encoder exists (nerf/run.py:227-229 leaves
        # embeddirs_fn None), rays have 8 columns; the fused pass has a variant without the view branch (and its backward).
        if need_time or any(net.input_ch != ef.out_dim for net in real):
            return None
        return ef.multires, 0, 0
    if not (isinstance(edf, EmbedFn) and edf.input_dims == 3):
        return None
    if need_time and not (isinstance(etf, EmbedFn) and etf.input_dims == 1):
        return None
    Lt = etf.multires if need_time else 0
    for net in nets:
        if net is None:
            continue
        if not isinstance(net, (vallina_NeRF, NeRFOriginal, DirectTemporalNeRF)) or not net._is_fused_arch():
            return None          # foreign modules and shapes the fused kernel is not built for: the op path
        if net.input_ch != ef.out_dim or net.input_ch_views != edf.out_dim:
            return None
        if isinstance(net, DirectTemporalNeRF) and (not need_time or net.input_ch_time != etf.out_dim):
            return None
    return ef.multires, edf.multires, Lt


# Arithmetic of the fused INFERENCE pass of the static nets: "fp32" (fp32 MFMA: the parity path, the default and what
# bench.py's headline measures), "bf16x3" (split-bf16 MFMA with fp32 accumulation, ~16 significant bits per operand; see
# include/swnerf.h swnerf_render_pass_x3), "bf16x3-fine" (below) or "bf16" (plain bf16 operands, a yardstick).
# Opt-in: set_precision() or SWNERF_PRECISION.
# "bf16x3-fine": fp32 for a pass that feeds the hierarchical resampling (the inverse CDF is a discontinuous function of the
# coarse weights, so last-bit differences there move fine samples), bf16x3 for every other pass: the depths, rgb0 and z_std
# are then exactly the fp32 path's and the image differs by the MLP rounding alone.
_PRECISIONS = {"fp32": 0, "bf16x3": 3, "bf16x3-fine": 3, "bf16": 1}
_WARNED = {}
PRECISION = os.environ.get("SWNERF_PRECISION", "fp32")
if PRECISION not in _PRECISIONS:
    raise ValueError(f"swnerf: SWNERF_PRECISION={PRECISION!r} is not one of {sorted(_PRECISIONS)}")


def set_precision(name):
    """Select the arithmetic of the fused inference pass; returns the previous setting."""
    global PRECISION
    if name not in _PRECISIONS:
        raise ValueError(f"swnerf.render.set_precision: unknown precision {name!r} (one of {sorted(_PRECISIONS)})")
    prev, PRECISION = PRECISION, name
    return prev


def prepack(*nets):
    """Build (or refresh) the packed weight streams the fused pass will ask for, on the CURRENT stream - for callers that
    then launch on several streams at once (parallel.frame_renderer): the cached blob must not be written on one stream while
    another reads it."""
    for net in nets:
        if net is None or not hasattr(net, "packed"):
            continue
        if isinstance(net, vallina_NeRF) and net._noview_params() is not None:
            net.packed_noview()
        elif net._is_fused_arch():
            net.packed()
            if _PRECISIONS[PRECISION]:
                net.packed_x3()


def render_pass(ray_batch, net, n_samples, *, z_vals=None, lindisp=False, t_rand=None, noise=None, white_bkgd=False,
                want=("rgb_map", "disp_map", "acc_map"), n_importance=0, u=None, run_deform=True, precision=None):
    """One launch of `swnerf_render_pass` (include/swnerf.h).  Returns a dict of the requested
    outputs among rgb_map disp_map acc_map depth_map weights raw dx z_out, plus z_fine/z_std
    when n_importance > 0.  precision: None = the module setting (PRECISION)."""
    noview = isinstance(net, vallina_NeRF) and net._noview_params() is not None
    out_ch = 4
    if noview:                                           # the fp32 pass without the view branch (SWNERF_NET_NOVIEW)
        packed, Lp, out_ch = net.packed_noview()
        kind, Ld, Lt, terms = _lib.NET_NOVIEW, 0, 0, 0
        if _PRECISIONS[PRECISION if precision is None else precision] and not _WARNED.get("noview_x3"):
            _WARNED["noview_x3"] = True
            import warnings
            warnings.warn(f"swnerf: precision {PRECISION if precision is None else precision!r} was requested, but the bf16 paths are built for the "
                          "nets WITH view directions only - this use_viewdirs=False net renders in fp32 (label your numbers accordingly)")
    else:
        kind, packed, Lp, Ld, Lt = net.packed()
        mode = PRECISION if precision is None else precision
        terms = _PRECISIONS[mode]
        if mode == "bf16x3-fine" and n_importance > 0:
            terms = 0
        if terms:
            packed, _, _ = net.packed_x3()
    rb = _lib.dev_f32(ray_batch, "ray_batch")
    N, cols = rb.shape
    S = int(n_samples)
    dev = rb.device
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    shapes = {"rgb_map": (N, 3), "disp_map": (N,), "acc_map": (N,), "depth_map": (N,), "weights": (N, S),
              "raw": (N, S, out_ch), "dx": (N, S, 3), "z_out": (N, S)}
    out = {k: new(*shapes[k]) for k in want}
    a = _lib.PassArgs()
    a.ray_batch, a.n_rays, a.cols, a.kind, a.packed = rb.data_ptr(), N, cols, kind, packed.data_ptr()
    a.run_deform, a.L_pos, a.L_dir, a.L_time, a.n_samples = int(bool(run_deform)), Lp, Ld, Lt, S
    a.out_ch = out_ch
    keep = [rb, packed]
    for name, t, last in (("z_vals", z_vals, S), ("t_rand", t_rand, S), ("noise", noise, S), ("u", u, int(n_importance))):
        if t is not None:
            t = _lib.dev_f32(t, name, last)
            if t.shape[0] != N:
                raise ValueError(f"swnerf.render_pass: {name} must have {N} rows, got {tuple(t.shape)}")
            keep.append(t)
            setattr(a, name, t.data_ptr())
    a.lindisp, a.white_bkgd = int(bool(lindisp)), int(bool(white_bkgd))
    for k, t in out.items():
        setattr(a, k, t.data_ptr())
    a.n_importance = int(n_importance)
    if n_importance > 0:
        out["z_fine"] = new(N, S + int(n_importance))
        out["z_std"] = new(N)
        a.z_fine, a.z_std = out["z_fine"].data_ptr(), out["z_std"].data_ptr()
    if PASS_HOOK is not None:
        PASS_HOOK("begin", N, S)
    if terms:
        _lib.check(_lib.lib().swnerf_render_pass_x3(a, terms, _lib.stream_of(rb)), "render_pass_x3")
    else:
        _lib.check(_lib.lib().swnerf_render_pass(a, _lib.stream_of(rb)), "render_pass")
    if PASS_HOOK is not None:
        PASS_HOOK("end", N, S)
    return out


class _FusedPassTrain(torch.autograd.Function):
    """The fused render pass under autograd (the reference's step is render -> img2mse ->
    loss.backward(), nerf/run.py:684-708).  forward = swnerf_render_pass_train: the inference pass that also saves
    activations, ReLU masks, the encodings and raw; backward = swnerf_render_pass_backward (one wave per ray: compositing
    backward in LDS, then the dX chain per tile) + one TN MFMA GEMM per Linear layer.  No [M,90] embedding, no cat, no
    pts tensor ever reaches HBM.  Gradients flow to the net's parameters only (rays are data; the depths of the fine
    pass are detached in the reference, nerf/run.py:398)."""

    @staticmethod
    def forward(ctx, net, rb, z_vals, S, lindisp, t_rand, noise, white_bkgd, n_importance, u, *params):
        noview = net._noview_params() is not None              # use_viewdirs=False: 8-column rays, raw [N,S,out_ch]
        out_ch = 4
        if noview:
            packed, Lp, out_ch = net.packed_noview()
            kind, Ld = _lib.NET_NOVIEW, 0
        else:
            kind, packed, Lp, Ld, _ = net.packed()
        L = _lib.lib()
        N, cols = rb.shape
        dev = rb.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        rows = L.swnerf_train_rows(N, S)
        act, bits, xs = new(rows, L.swnerf_act_floats_per_row()), new(L.swnerf_mask_floats(rows)), new(rows, L.swnerf_xs_floats_per_row())
        raw, rgb, disp, acc = new(N, S, out_ch), new(N, 3), new(N), new(N)
        a = _lib.PassArgs()
        a.ray_batch, a.n_rays, a.cols, a.kind, a.packed = rb.data_ptr(), N, cols, kind, packed.data_ptr()
        a.out_ch = out_ch
        a.run_deform, a.L_pos, a.L_dir, a.L_time, a.n_samples = 0, Lp, Ld, 0, S
        a.lindisp, a.white_bkgd = int(bool(lindisp)), int(bool(white_bkgd))
        a.rgb_map, a.disp_map, a.acc_map, a.raw = rgb.data_ptr(), disp.data_ptr(), acc.data_ptr(), raw.data_ptr()
        if z_vals is not None:
            z = z_vals
            a.z_vals = z.data_ptr()
        else:
            z = new(N, S)
            a.z_out = z.data_ptr()
        for name, t in (("t_rand", t_rand), ("noise", noise), ("u", u)):
            if t is not None:
                setattr(a, name, t.data_ptr())
        a.n_importance = int(n_importance)
        if n_importance > 0:
            z_fine, z_std = new(N, S + int(n_importance)), new(N)
            a.z_fine, a.z_std = z_fine.data_ptr(), z_std.data_ptr()
        else:
            z_fine, z_std = new(0), new(0)
        if PASS_HOOK is not None:
            PASS_HOOK("begin", N, S)
        _lib.check(L.swnerf_render_pass_train(a, _lib.ptr(act), _lib.ptr(bits), _lib.ptr(xs), _lib.stream_of(rb)), "render_pass_train")
        if PASS_HOOK is not None:
            PASS_HOOK("end", N, S)
        ctx.net, ctx.S, ctx.white, ctx.bands = net, S, bool(white_bkgd), (Lp, Ld)
        ctx.has_noise, ctx.noview, ctx.out_ch = noise is not None, noview, out_ch
        ctx.save_for_backward(rb, z, raw, act, bits, xs, noise if noise is not None else new(0), *params)
        ctx.mark_non_differentiable(z_fine, z_std)
        ctx.set_materialize_grads(False)       # an output the loss does not use arrives as None (no zero fill, no read of zeros in the kernel)
        # raw is an output as well (retraw=True is what the reference's train() passes, nerf/run.py:685): a gradient
        # arriving on it is added to d raw in the backward kernel
        return rgb, disp, acc, z_fine, z_std, raw

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, _gz, _gs, g_raw):
        from .wgrad import WeightGrads, _Fan, _chunk_gemms, NARROW_FUSED, NOVIEW_NARROW_FUSED
        rb, z, raw, act, bits, xs, noise, *params = ctx.saved_tensors
        net, S = ctx.net, ctx.S
        Lp, Ld = ctx.bands
        L = _lib.lib()
        N, cols = rb.shape
        st = _lib.stream_of(rb)
        c = lambda g: None if g is None else g.contiguous().float()
        g_rgb, g_disp, g_acc, g_raw = c(g_rgb), c(g_disp), c(g_acc), c(g_raw)
        nv = ctx.noview
        wg = WeightGrads(L, "noview" if nv else "canon", params, fused=True, Cpos=net.input_ch, Cdir=0 if nv else net.input_ch_views, bands=(Lp, Ld, 0))
        # The gradient buffer [rows, 2432] is as large as the saved activations; the dX chain and the GEMMs that consume
        # it run per CHUNK of rays, so only one chunk of it is ever alive (GEMMs accumulate: C += A^T.B).  A chunk is
        # 393 216 rows at the C2 shape - large enough for the split-K GEMMs to fill the chip.
        rows_per_ray = act.shape[0] // N
        chunk = max(4, (TRAIN_BWD_CHUNK_ROWS // rows_per_ray) // 4 * 4)
        packed_bwd = net.packed_bwd_noview() if nv else net.packed_bwd()
        mask_per_ray = bits.numel() // N
        sl = lambda t, r0, r1: None if t is None else t[r0:r1]
        grad = torch.empty((min(N, chunk) * rows_per_ray, act.shape[1]), dtype=torch.float32, device=rb.device)
        d_raw = torch.empty((min(N, chunk) * rows_per_ray, 8 if nv else 4), dtype=torch.float32, device=rb.device)
        fan = _Fan(rb.device)                                    # the GEMMs of a chunk fan out over side streams (wgrad._Fan)
        for r0 in range(0, N, chunk):
            r1 = min(N, r0 + chunk)
            n, m = r1 - r0, (r1 - r0) * rows_per_ray
            common = (_lib.ptr(packed_bwd), _lib.ptr(bits[r0 * mask_per_ray:r1 * mask_per_ray]), _lib.ptr(raw[r0:r1]), _lib.ptr(z[r0:r1]),
                      _lib.ptr(rb[r0:r1]), cols, _lib.ptr(noise[r0:r1]) if ctx.has_noise else None, n, S, int(ctx.white))
            grads_in = (_lib.ptr(sl(g_rgb, r0, r1)), _lib.ptr(sl(g_disp, r0, r1)), _lib.ptr(sl(g_acc, r0, r1)), _lib.ptr(sl(g_raw, r0, r1)),
                        _lib.ptr(grad), _lib.ptr(d_raw), st)
            if nv:
                _lib.check(L.swnerf_render_pass_backward_noview(*common, ctx.out_ch, *grads_in), "render_pass_backward_noview")
            else:
                _lib.check(L.swnerf_render_pass_backward(*common, *grads_in), "render_pass_backward")
            a0, a1 = r0 * rows_per_ray, r1 * rows_per_ray
            job = lambda st_, part: wg.chunk(st_, m, grad[:m], act[a0:a1], xs[a0:a1], d_raw[:m], part=part)
            _chunk_gemms(L, fan, m, [job], rest_on_main=NARROW_FUSED and (not nv or NOVIEW_NARROW_FUSED))
        return (None,) * 10 + tuple(gi.to(p.dtype) for gi, p in zip(wg.finish(st), params))


def render_pass_train(ray_batch, net, n_samples, *, z_vals=None, lindisp=False, t_rand=None, noise=None, white_bkgd=False,
                      n_importance=0, u=None):
    """One differentiable fused pass (`_FusedPassTrain`): dict with rgb_map disp_map acc_map raw (+ z_fine z_std)."""
    from .model import _CANON_ORDER, _NOVIEW_ORDER
    rb = _lib.dev_f32(ray_batch.detach(), "ray_batch")
    N, S = rb.shape[0], int(n_samples)
    chk = lambda t, name, last: None if t is None else _lib.dev_f32(t.detach(), name, last)
    z_vals, t_rand, noise, u = chk(z_vals, "z_vals", S), chk(t_rand, "t_rand", S), chk(noise, "noise", S), chk(u, "u", int(n_importance))
    for name, t in (("z_vals", z_vals), ("t_rand", t_rand), ("noise", noise), ("u", u)):
        if t is not None and t.shape[0] != N:
            raise ValueError(f"swnerf.render_pass_train: {name} must have {N} rows, got {tuple(t.shape)}")
    sd = dict(net.named_parameters())
    rgb, disp, acc, z_fine, z_std, raw = _FusedPassTrain.apply(net, rb, z_vals, S, bool(lindisp), t_rand, noise, bool(white_bkgd),
                                                               int(n_importance), u,
                                                               *[sd[n] for n in (_NOVIEW_ORDER if net._noview_params() is not None else _CANON_ORDER)])
    out = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "raw": raw}
    if n_importance > 0:
        out["z_fine"], out["z_std"] = z_fine, z_std
    return out


PASS_MAX_COARSE, PASS_MAX_SORT = 256, 1024   # csrc/render_pass.h SW_LDS_SC / SW_LDS_SORT: the in-LDS resampling of swnerf_render_pass


def pass_can_resample(N_samples, N_importance):
    """Whether the fused coarse pass can do the hierarchical resampling of these counts in its LDS slice."""
    return 3 <= N_samples <= PASS_MAX_COARSE and N_samples + N_importance <= PASS_MAX_SORT


def resample_ops(z_vals, weights, N_importance, u=None):
    """nerf/run.py:394-400,416 on the individual ops, for sample counts beyond the fused pass's LDS slice (the reference
    takes any N_samples): sample_pdf on the mid-points (the HIP op; device tensor ops past 1024 bins), torch.sort of
    cat[z_vals, z_samples], std of the samples.  u None = deterministic (perturb == 0).  -> z_fine [N, S+Ni], z_std [N]"""
    if z_vals.shape[-1] < 3:
        # weights[..., 1:-1] is empty: the reference's sample_pdf indexes an empty cdf with -1 and raises (ray.py:113-146)
        raise ValueError(f"swnerf.render_rays: hierarchical resampling needs N_samples >= 3 (got {z_vals.shape[-1]}); the reference fails there too")
    z_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, det=(u is None), u=u).detach()
    z_fine, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
    return z_fine.contiguous(), torch.std(z_samples, dim=-1, unbiased=False)


def coarse_pass_resampled(ray_batch, net, N_samples, N_importance, *, want, u=None, **kw):
    """The coarse pass + resampling of render_rays: ONE fused launch when the counts fit its LDS slice, otherwise the
    fused pass without resampling (weights and depths written out) followed by resample_ops.  Returns render_pass's dict
    with z_fine / z_std either way."""
    if pass_can_resample(N_samples, N_importance):
        return render_pass(ray_batch, net, N_samples, want=want, n_importance=N_importance, u=u, **kw)
    p0 = render_pass(ray_batch, net, N_samples, want=list(want) + ["weights", "z_out"], **kw)
    p0["z_fine"], p0["z_std"] = resample_ops(p0["z_out"], p0["weights"], N_importance, u)
    return p0


TRAIN_FUSED_MAX_SAMPLES = 256      # include/swnerf.h: swnerf_render_pass_train
# rows of the gradient buffer alive at once in the fused backward (393 216 rows = 3.8 GB); each chunk costs a backward-kernel
# tail, one grouped GEMM launch and the narrow GEMMs' tail, so not smaller than needed
TRAIN_BWD_CHUNK_ROWS = int(os.environ.get("SWNERF_TRAIN_BWD_CHUNK_ROWS", "393216"))


def _rng_inputs(N, N_samples, N_importance, perturb, raw_noise_std, pytest, dev):
    """The three random tensors of render_rays (nerf/run.py:375-381, ray.py:117-132, :176-184)."""
    t_rand = u = None
    if perturb > 0.:
        t_rand = torch.rand((N, N_samples), device=dev)
        if pytest:
            np.random.seed(0)
            t_rand = torch.Tensor(np.random.rand(N, N_samples)).to(dev)
        if N_importance > 0:
            u = torch.rand((N, N_importance), device=dev)
            if pytest:
                np.random.seed(0)
                u = torch.Tensor(np.random.rand(N, N_importance)).to(dev)

    def noise(S):
        if not raw_noise_std > 0.:
            return None
        if pytest:
            np.random.seed(0)
            return torch.Tensor(np.random.rand(N, S) * raw_noise_std).to(dev)
        return torch.randn((N, S), device=dev) * raw_noise_std
    return t_rand, u, noise


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False):
    """nerf/run.py:316-422."""
    plan = fused_plan(network_query_fn, [network_fn, network_fine], allow_train=True) if ray_batch.shape[-1] in (8, 11) else None
    if plan is not None and nets_without_views([network_fn, network_fine]) != (ray_batch.shape[-1] == 8):
        plan = None                                      # 8 columns <=> nets without view directions
    training = wants_grad([network_fn, network_fine])
    if plan is not None and training:
        # the fused pass has a backward for the static nets, up to 256 samples per pass; anything else trains on the
        # differentiable op path
        ok = (ray_batch.shape[0] > 0 and N_samples <= TRAIN_FUSED_MAX_SAMPLES
              and N_samples + max(0, N_importance) <= TRAIN_FUSED_MAX_SAMPLES and os.environ.get("SWNERF_TRAIN_OP_PATH") != "1"
              and all(net is None or isinstance(net, vallina_NeRF) for net in (network_fn, network_fine)))
        if not ok:
            plan = None
    if plan is None:
        return _render_rays_unfused(ray_batch, network_fn, network_query_fn, N_samples, retraw, lindisp, perturb,
                                    N_importance, network_fine, white_bkgd, raw_noise_std, pytest)
    N = ray_batch.shape[0]
    t_rand, u, noise = _rng_inputs(N, N_samples, N_importance, perturb, raw_noise_std, pytest, ray_batch.device)
    if training:
        p0 = render_pass_train(ray_batch, network_fn, N_samples, lindisp=lindisp, t_rand=t_rand, noise=noise(N_samples),
                               white_bkgd=white_bkgd, n_importance=max(0, N_importance), u=u)
        if N_importance <= 0:
            ret = {'rgb_map': p0["rgb_map"], 'disp_map': p0["disp_map"], 'acc_map': p0["acc_map"]}
            if retraw:
                ret['raw'] = p0["raw"]
            return ret
        S1 = N_samples + N_importance
        run_fn = network_fn if network_fine is None else network_fine
        p1 = render_pass_train(ray_batch, run_fn, S1, z_vals=p0["z_fine"], noise=noise(S1), white_bkgd=white_bkgd)
        ret = {'rgb_map': p1["rgb_map"], 'disp_map': p1["disp_map"], 'acc_map': p1["acc_map"]}
        if retraw:
            ret['raw'] = p1["raw"]
        ret.update({'rgb0': p0["rgb_map"], 'disp0': p0["disp_map"], 'acc0': p0["acc_map"], 'z_std': p0["z_std"]})
        return ret
    want = ["rgb_map", "disp_map", "acc_map"] + (["raw"] if (retraw and N_importance <= 0) else [])
    if N_importance <= 0:
        p0 = render_pass(ray_batch, network_fn, N_samples, lindisp=lindisp, t_rand=t_rand, noise=noise(N_samples),
                         white_bkgd=white_bkgd, want=want)
        ret = {'rgb_map': p0["rgb_map"], 'disp_map': p0["disp_map"], 'acc_map': p0["acc_map"]}
        if retraw:
            ret['raw'] = p0["raw"]
        return ret
    p0 = coarse_pass_resampled(ray_batch, network_fn, N_samples, N_importance, want=want, u=u, lindisp=lindisp, t_rand=t_rand,
                               noise=noise(N_samples), white_bkgd=white_bkgd)
    S1 = N_samples + N_importance
    if Z_TAP is not None:
        Z_TAP["fused"] = p0["z_fine"]
    run_fn = network_fn if network_fine is None else network_fine
    p1 = render_pass(ray_batch, run_fn, S1, z_vals=p0["z_fine"], noise=noise(S1), white_bkgd=white_bkgd,
                     want=["rgb_map", "disp_map", "acc_map"] + (["raw"] if retraw else []))
    ret = {'rgb_map': p1["rgb_map"], 'disp_map': p1["disp_map"], 'acc_map': p1["acc_map"]}
    if retraw:
        ret['raw'] = p1["raw"]
    ret['rgb0'], ret['disp0'], ret['acc0'] = p0["rgb_map"], p0["disp_map"], p0["acc_map"]
    ret['z_std'] = p0["z_std"]
    return ret


def sample_coarse(ray_batch, N_samples, lindisp=False, t_rand=None, want_pts=False):
    """The coarse sampling of render_rays (nerf/run.py:355-385) as one HIP op: z_vals [N,S] (and pts [N,S,3])."""
    rb = _lib.dev_f32(ray_batch, "ray_batch")
    N, cols = rb.shape
    z = torch.empty((N, int(N_samples)), dtype=torch.float32, device=rb.device)
    pts = torch.empty((N, int(N_samples), 3), dtype=torch.float32, device=rb.device) if want_pts else None
    tr = None if t_rand is None else _lib.dev_f32(t_rand, "t_rand", int(N_samples))
    _lib.check(_lib.lib().swnerf_sample_coarse(_lib.ptr(rb), N, cols, int(N_samples), int(bool(lindisp)), _lib.ptr(tr),
                                               _lib.ptr(z), _lib.ptr(pts), _lib.stream_of(rb)), "sample_coarse")
    return (z, pts) if want_pts else z


def _coarse_z(near, far, N_rays, N_samples, lindisp, perturb, pytest, ray_batch=None):
    """nerf/run.py:361-383 (unfused path only): the HIP op when the ray batch is at hand, else torch ops."""
    dev = near.device
    t_rand = None
    if perturb > 0.:
        t_rand = torch.rand((N_rays, N_samples), device=dev)
        if pytest:
            np.random.seed(0)
            t_rand = torch.Tensor(np.random.rand(N_rays, N_samples)).to(dev)
    if ray_batch is not None and ray_batch.is_cuda:
        return sample_coarse(ray_batch.detach(), N_samples, lindisp, t_rand)
    t_vals = torch.linspace(0., 1., steps=N_samples, device=dev)
    z_vals = near * (1. - t_vals) + far * t_vals if not lindisp else 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
    z_vals = z_vals.expand([N_rays, N_samples])
    if t_rand is not None:
        mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        z_vals = lower + (upper - lower) * t_rand
    return z_vals


def _render_rays_unfused(ray_batch, network_fn, network_query_fn, N_samples, retraw, lindisp, perturb, N_importance,
                         network_fine, white_bkgd, raw_noise_std, pytest):
    """The reference's op sequence (nerf/run.py:354-416) on the individual HIP ops; taken when the
    nets / encoders are not the ones the fused pass is built for."""
    N_rays = ray_batch.shape[0]
    rays_o, rays_d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    viewdirs = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    bounds = torch.reshape(ray_batch[..., 6:8], [-1, 1, 2])
    near, far = bounds[..., 0], bounds[..., 1]
    z_vals = _coarse_z(near, far, N_rays, N_samples, lindisp, perturb, pytest, ray_batch)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    raw = network_query_fn(pts, viewdirs, network_fn)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
        z_vals_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = sample_pdf(z_vals_mid, weights[..., 1:-1], N_importance, det=(perturb == 0.), pytest=pytest).detach()
        z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
        if Z_TAP is not None:
            Z_TAP["unfused"] = z_vals
        pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
        run_fn = network_fn if network_fine is None else network_fine
        raw = network_query_fn(pts, viewdirs, run_fn)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest=pytest)
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'], ret['disp0'], ret['acc0'] = rgb_map_0, disp_map_0, acc_map_0
        ret['z_std'] = torch.std(z_samples, dim=-1, unbiased=False)
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """nerf/run.py:90-102.  (The fused pass needs no chunking for memory - activations never
    reach HBM - but `chunk` is honoured so results are laid out the same way.)"""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}


def pack_ray_batch(rays_o, rays_d, near, far, frame_time=None, ndc=False, H=0, W=0, focal=1.):
    """rows [o d near far (t) viewdirs] - nerf/run.py:137-158 / d_nerf/run_dnerf.py:137-160."""
    rays_d = _lib.dev_f32(torch.reshape(rays_d, [-1, 3]), "rays_d", 3)
    rays_o = _lib.dev_f32(torch.reshape(rays_o.expand(rays_d.shape) if rays_o.numel() != rays_d.numel() else rays_o, [-1, 3]), "rays_o", 3)
    n = rays_d.shape[0]
    scalar = lambda v: not isinstance(v, (torch.Tensor, np.ndarray))
    cols = 12 if frame_time is not None else 11
    out = torch.empty((n, cols), dtype=torch.float32, device=rays_d.device)
    _lib.check(_lib.lib().swnerf_pack_ray_batch(
        _lib.ptr(rays_o), _lib.ptr(rays_d), n, float(near) if scalar(near) else 0., float(far) if scalar(far) else 0.,
        int(frame_time is not None), float(frame_time) if (frame_time is not None and scalar(frame_time)) else 0.,
        int(bool(ndc)), int(H), int(W), float(focal), _lib.ptr(out), _lib.stream_of(out)), "pack_ray_batch")
    for col, v in ((6, near), (7, far), (8, frame_time)):       # per-ray arrays (render() docstring) via torch
        if v is not None and not scalar(v):
            out[:, col] = torch.as_tensor(v, dtype=torch.float32, device=out.device).reshape(-1)
    return out


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """nerf/run.py:105-169 -> [rgb_map, disp_map, acc_map, extras]."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
    viewsrc = rays_d
    if c2w_staticcam is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
    sh = rays_d.shape
    if c2w_staticcam is not None:
        # view directions from c2w, geometry from the static camera (nerf/run.py:139-142)
        vb = pack_ray_batch(rays_o, viewsrc, near, far)
        rb = pack_ray_batch(rays_o, rays_d, near, far, ndc=ndc, H=H, W=W, focal=K[0][0])
        rb[:, -3:] = vb[:, -3:]
    else:
        rb = pack_ray_batch(rays_o, rays_d, near, far, ndc=ndc, H=H, W=W, focal=K[0][0])
    if not use_viewdirs:
        rb = rb[:, :8].contiguous()          # rays = cat[o, d, near, far] without view directions (nerf/run.py:152-157)
    all_ret = batchify_rays(rb, chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]


def pipelined_frames(frames, consume):
    """The host side of render_path, one frame deep: `frames` yields (i, rgb, disp) device tensors as each frame's kernels are
    ENQUEUED; the frame's pixels go to pinned host memory asynchronously, and `consume(i, rgb_np, disp_np)` - PSNR print, PNG
    encoding, list append: tens of ms of CPU work per 800x800 frame - runs for frame i-1 while the GPU renders frame i.  (The
    reference does `rgb.cpu().numpy()` + imageio.imwrite between frames, nerf/run.py:199-213, with the GPU idle meanwhile.)
    Order and values are the reference's; CPU tensors pass straight through."""
    pending = None
    staging = {}                                     # two page-locked staging pairs, reused ping-pong (never one per frame)

    def done(item):
        i, h_rgb, h_disp, ev = item
        if ev is not None:
            ev.synchronize()
            # hand out PAGEABLE copies: the caller keeps every frame (render_path stacks them at the end) and a 200-frame
            # 800x800 path would otherwise hold ~2 GB of page-locked memory; the staging pair is free for frame i+2
            consume(i, np.array(h_rgb.numpy()), np.array(h_disp.numpy()))
        else:
            consume(i, h_rgb.numpy(), h_disp.numpy())

    k = 0
    for i, rgb, disp in frames:
        if rgb.is_cuda:
            key = (k & 1, tuple(rgb.shape), tuple(disp.shape), rgb.dtype, disp.dtype)
            if key not in staging:
                for old_key in [q for q in staging if q[0] == (k & 1)]:
                    del staging[old_key]             # a path whose frame size changes: drop the slot's old pair
                staging[key] = (torch.empty(rgb.shape, dtype=rgb.dtype, pin_memory=True), torch.empty(disp.shape, dtype=disp.dtype, pin_memory=True))
            h_rgb, h_disp = staging[key]
            h_rgb.copy_(rgb.detach(), non_blocking=True)
            h_disp.copy_(disp.detach(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(rgb.device))
            k += 1
        else:
            h_rgb, h_disp, ev = rgb.detach(), disp.detach(), None
        if pending is not None:
            done(pending)
        pending = (i, h_rgb, h_disp, ev)
    if pending is not None:
        done(pending)


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """nerf/run.py:172-219; with `savedir` every frame is also written as '{:03d}.png' of to8b(rgb) (:210-213)."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps = [], []

    def frames():
        for i, c2w in enumerate(render_poses):
            rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
            yield i, rgb, disp

    def consume(i, rgb, disp):
        rgbs.append(rgb)
        disps.append(disp)
        if gt_imgs is not None and render_factor == 0:                       # nerf/run.py:204-206: PSNR against the ground truth
            gt = gt_imgs[i]
            gt = gt.cpu().numpy() if isinstance(gt, torch.Tensor) else np.asarray(gt)
            print(-10. * np.log10(np.mean(np.square(rgb - gt))))
        if savedir is not None:
            write_png(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgb))
    pipelined_frames(frames(), consume)
    return np.stack(rgbs, 0), np.stack(disps, 0)
