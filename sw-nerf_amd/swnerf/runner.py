"""`create_nerf` of the two runners (nerf/run.py:222-313, d_nerf/run_dnerf.py:238-352): what stands immediately before
the render path - embedders, the network(s), the `network_query_fn` closure, Adam, checkpoint reload - and returns
(render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer) exactly as the reference does, so a `train()`
written against the reference only swaps its imports.  `args` is any object with the reference's option names
(utils.py config_parser / run_dnerf.py config_parser); the closure is built inside the function, which is how
`render.fused_plan` finds the encoders and sends `render_rays` to the fused HIP pass."""
import torch

from . import render, render_dnerf
from .checkpoint import reload_latest
from .embedder import get_embedder
from .model import vallina_NeRF, NeRF


def _render_kwargs(args, network_query_fn, model, model_fine, extra=None):
    kw = {
        'network_query_fn': network_query_fn,
        'perturb': args.perturb,
        'N_importance': args.N_importance,
        'network_fine': model_fine,
        'N_samples': args.N_samples,
        'network_fn': model,
        'use_viewdirs': args.use_viewdirs,
        'white_bkgd': args.white_bkgd,
        'raw_noise_std': args.raw_noise_std,
    }
    kw.update(extra or {})
    if args.dataset_type != 'llff' or args.no_ndc:               # nerf/run.py:296-299
        kw['ndc'] = False
        kw['lindisp'] = args.lindisp
    test = dict(kw)
    test['perturb'] = False                                      # nerf/run.py:301-303
    test['raw_noise_std'] = 0.
    return kw, test


def _device(device):
    if device is not None:
        return torch.device(device)
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def create_nerf(args, device=None):
    """nerf/run.py:222-313 (static NeRF: coarse `vallina_NeRF` + fine one when N_importance > 0)."""
    device = _device(device)
    embed_fn, input_ch = get_embedder(args.multires, input_dims=3, i=args.i_embed)
    input_ch_views, embeddirs_fn = 0, None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, input_dims=3, i=args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = vallina_NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                         input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = vallina_NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                                  skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
        grad_vars += list(model_fine.parameters())
    netchunk = args.netchunk
    network_query_fn = lambda inputs, viewdirs, network_fn: render.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=netchunk)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start, _ = reload_latest(args.basedir, args.expname, model, model_fine, optimizer, ft_path=args.ft_path,
                             no_reload=args.no_reload, map_location=device)
    train, test = _render_kwargs(args, network_query_fn, model, model_fine)
    return train, test, start, grad_vars, optimizer


def create_dnerf(args, device=None):
    """d_nerf/run_dnerf.py:238-352 (`create_nerf` of the D-NeRF runner): `NeRF.get_by_name(args.nerf_type, ...)`,
    the time encoder, `use_two_models_for_fine`.  fp32 only: `do_half_precision` (apex amp) is refused."""
    if getattr(args, "do_half_precision", False):
        raise NotImplementedError("swnerf.create_dnerf: do_half_precision (apex amp) is not built; the HIP path is fp32")
    device = _device(device)
    embed_fn, input_ch = get_embedder(args.multires, 3, args.i_embed)
    embedtime_fn, input_ch_time = get_embedder(args.multires, 1, args.i_embed)
    input_ch_views, embeddirs_fn = 0, None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, 3, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    make = lambda D, W: NeRF.get_by_name(args.nerf_type, D=D, W=W, input_ch=input_ch, output_ch=output_ch, skips=skips,
                                         input_ch_views=input_ch_views, input_ch_time=input_ch_time,
                                         use_viewdirs=args.use_viewdirs, embed_fn=embed_fn,
                                         zero_canonical=not args.not_zero_canonical).to(device)
    model = make(args.netdepth, args.netwidth)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.use_two_models_for_fine:
        model_fine = make(args.netdepth_fine, args.netwidth_fine)
        grad_vars += list(model_fine.parameters())
    netchunk, discr = args.netchunk, args.nerf_type != "temporal"
    network_query_fn = lambda inputs, viewdirs, ts, network_fn: render_dnerf.run_network(
        inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn,
        netchunk=netchunk, embd_time_discr=discr)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start, _ = reload_latest(args.basedir, args.expname, model, model_fine, optimizer, ft_path=args.ft_path,
                             no_reload=args.no_reload, map_location=device)
    train, test = _render_kwargs(args, network_query_fn, model, model_fine,
                                 {'use_two_models_for_fine': args.use_two_models_for_fine})
    return train, test, start, grad_vars, optimizer
