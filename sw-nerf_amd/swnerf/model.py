"""Drop-in for the reference's model.py: vallina_NeRF, NeRFOriginal, DirectTemporalNeRF and the
NeRF.get_by_name factory as nn.Modules with the SAME parameter names and shapes (so the
reference's checkpoints `network_fn_state_dict` / `network_fine_state_dict` load unchanged,
nerf/run.py:269-280), whose forward runs the register-resident MFMA kernel
(csrc/mlp_core.h) through `swnerf_mlp_forward`.

The register-resident kernel is built for the configuration every shipped config uses: D=8, W=256, skips=[4],
use_viewdirs=True with get_embedder-sized inputs.  Any other shape (use_viewdirs=False - the reference's argparse
default -, other D / W / skips / input sizes) runs layer by layer on the generic MFMA GEMM kernels (swnerf/generic.py,
csrc/generic_kernels.hip): slower, same results, differentiable.
TNeRF (model.py:152-210) is out of scope.
Training: with grad enabled, forward saves the activations and backward runs the
register-resident dX chain + TN MFMA GEMMs (`_MlpTrain`, `_DnerfTrain`); gradients w.r.t. the embedded inputs are
not produced (rays are data in the reference's train()).
"""
import ctypes
import os
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401
import numpy as np

from . import _lib
from .embedder import img2mse, mse2psnr, to8b  # noqa: F401

_CANON_ORDER = ([f"pts_linears.{i}.{p}" for i in range(8) for p in ("weight", "bias")]
                + [f"{n}.{p}" for n in ("views_linears.0", "feature_linear", "alpha_linear", "rgb_linear")
                   for p in ("weight", "bias")])
_NOVIEW_ORDER = ([f"pts_linears.{i}.{p}" for i in range(8) for p in ("weight", "bias")]
                 + [f"output_linear.{p}" for p in ("weight", "bias")])
_DEFORM_ORDER = ([f"_time.{i}.{p}" for i in range(8) for p in ("weight", "bias")]
                 + [f"_time_out.{p}" for p in ("weight", "bias")])


def _bands(ch, d):
    """number of frequency bands L with ch == d*(1+2L), or None"""
    if ch % d:
        return None
    q = ch // d - 1
    return q // 2 if q >= 0 and q % 2 == 0 else None


class _NoBackward(torch.autograd.Function):
    """Marks the output of an INFERENCE-kernel forward that was called with grad enabled on a path that has no backward (the
    D-NeRF t == 0 / no_grad helpers route around it; a caller who differentiates such an output gets a clear error instead of
    silently missing gradients)."""

    @staticmethod
    def forward(ctx, out, *params):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, *g):
        raise NotImplementedError(
            "swnerf: this output came from an inference-only kernel launch (module.forward under torch.no_grad() semantics); "
            "differentiate through render_rays / module.forward with grad enabled from the start - those run the training kernels")


# the weight-gradient machinery lives in wgrad.py; these names are part of what tests / tools reach through `model`
from .wgrad import WeightGrads, _Fan, _Group, _gemm_tn, _gemm_tn_fused, _chunk_gemms, NARROW_FUSED, SW_ACT_HV  # noqa: E402,F401


class _MlpTrain(torch.autograd.Function):
    """Differentiable forward of the static 8x256 net: the forward kernel saves
    every layer's activation; backward = the register-resident dX chain over the transposed weight
    stream + one TN MFMA GEMM per Linear layer for dW / db.  Inputs (embedded points) get no gradient,
    like in the reference's training loop (rays are data)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        kind, packed, Lp, Ld, _ = module.packed()
        L = _lib.lib()
        M = x.shape[0]
        out = torch.empty((M, 4), dtype=torch.float32, device=x.device)
        act = torch.empty((M, L.swnerf_act_floats_per_row()), dtype=torch.float32, device=x.device)
        bits = torch.empty(L.swnerf_mask_floats(M), dtype=torch.float32, device=x.device)
        _lib.check(L.swnerf_mlp_forward_train(_lib.ptr(packed), _lib.ptr(x), M, Lp, Ld, _lib.ptr(out), _lib.ptr(act),
                                              _lib.ptr(bits), _lib.stream_of(x)), "mlp_forward_train")
        ctx.module, ctx.bands = module, (Lp, Ld)
        ctx.save_for_backward(x, act, bits, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("swnerf: gradients w.r.t. the embedded inputs are not built (rays are data in train())")
        x, act, bits, *params = ctx.saved_tensors
        module = ctx.module
        L = _lib.lib()
        M = x.shape[0]
        d_out = d_out.contiguous().float()
        grad = torch.empty_like(act)
        st = _lib.stream_of(x)
        _lib.check(L.swnerf_mlp_backward_dx(_lib.ptr(module.packed_bwd()), _lib.ptr(bits), _lib.ptr(d_out), M, _lib.ptr(grad), st),
                   "mlp_backward_dx")
        wg = WeightGrads(L, "canon", params, fused=False, Cpos=module.input_ch, Cdir=module.input_ch_views, bands=ctx.bands + (0,))
        wg.chunk(st, M, grad, act, x, d_out)
        return (None, None) + tuple(gi.to(p.dtype) for gi, p in zip(wg.finish(st), params))


class _DnerfTrain(torch.autograd.Function):
    """Differentiable DirectTemporalNeRF.forward for t != 0 (model.py:128-151): returns (out, dx), both with
    gradients - d_nerf/run_dnerf.py:690-725 puts a TV loss on dx (`position_delta`) next to the image loss.
    Forward = deformation net (activations saved) -> gamma(x + dx) -> canonical net (activations saved).
    Backward = canonical dX chain incl. d gamma(x+dx) -> d(x+dx) through the sin/cos Jacobian, the deformation
    dX chain seeded with d dx = d(x+dx) + d position_delta, then one TN GEMM per Linear layer of both nets.
    params: the 24 `_occ` tensors (_CANON_ORDER) then the 18 `_time`/`_time_out` tensors (_DEFORM_ORDER)."""

    @staticmethod
    def forward(ctx, module, x, t_emb, *params):
        kind, packed, Lp, Ld, Lt = module.packed()
        occ = module._occ
        L = _lib.lib()
        M = x.shape[0]
        st = _lib.stream_of(x)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)
        nact = L.swnerf_act_floats_per_row()
        nbits = L.swnerf_mask_floats(M)
        dx, act_d, bits_d = new(M, 3), new(M, nact), new(nbits)
        _lib.check(L.swnerf_deform_forward_train(_lib.ptr(packed), _lib.ptr(x), _lib.ptr(t_emb), M, Lp, Ld, Lt,
                                                 _lib.ptr(dx), _lib.ptr(act_d), _lib.ptr(bits_d), st), "deform_forward_train")
        Cpos = module.input_ch
        pts2 = x[:, :3] + dx                                                     # model.py:147
        x2 = new(M, x.shape[1])
        x2[:, Cpos:] = x[:, Cpos:]
        emb = new(M, Cpos)
        _lib.check(L.swnerf_embed(_lib.ptr(pts2), M, 3, Lp, _lib.ptr(emb), st), "embed")   # model.py:148-149
        x2[:, :Cpos] = emb
        out, act_c, bits_c = new(M, 4), new(M, nact), new(nbits)
        _lib.check(L.swnerf_mlp_forward_train(_lib.ptr(occ.packed()[1]), _lib.ptr(x2), M, Lp, Ld, _lib.ptr(out), _lib.ptr(act_c),
                                              _lib.ptr(bits_c), st), "mlp_forward_train")
        ctx.module = module
        ctx.save_for_backward(x, t_emb, x2, pts2, act_d, act_c, bits_d, bits_c, *params)
        return out, dx

    @staticmethod
    def backward(ctx, d_out, d_dx):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise NotImplementedError("swnerf: gradients w.r.t. the embedded inputs are not built (rays are data in train())")
        x, t_emb, x2, pts2, act_d, act_c, bits_d, bits_c, *params = ctx.saved_tensors
        module = ctx.module
        occ = module._occ
        L = _lib.lib()
        M = x.shape[0]
        st = _lib.stream_of(x)
        kind, names, Lp, Ld, Lt = module._pack_params()
        Cpos, Cdir, Ct = module.input_ch, module.input_ch_views, module.input_ch_time
        d_out = (torch.zeros((M, 4), dtype=torch.float32, device=x.device) if d_out is None else d_out.contiguous().float())
        grad_c, d_pts = torch.empty_like(act_c), torch.empty((M, 3), dtype=torch.float32, device=x.device)
        _lib.check(L.swnerf_mlp_backward_dx_pts(_lib.ptr(occ.packed_bwd(_lib.BWD_CANON_INPUT_GRAD)), _lib.ptr(bits_c), _lib.ptr(d_out),
                                                _lib.ptr(pts2), M, Lp, _lib.ptr(grad_c), _lib.ptr(d_pts), st), "mlp_backward_dx_pts")
        wc = WeightGrads(L, "canon", params[:24], fused=False, Cpos=Cpos, Cdir=Cdir, bands=(Lp, Ld, Lt))
        wc.chunk(st, M, grad_c, act_c, x2, d_out)
        g_dx = d_pts if d_dx is None else (d_pts + d_dx.float()).contiguous()
        grad_d = torch.empty_like(act_d)
        _lib.check(L.swnerf_deform_backward_dx(_lib.ptr(module.packed_bwd(_lib.BWD_DEFORM)), _lib.ptr(bits_d), _lib.ptr(g_dx), M,
                                               _lib.ptr(grad_d), st), "deform_backward_dx")
        wd = WeightGrads(L, "deform", params[24:], fused=False, Cpos=Cpos, Ct=Ct, bands=(Lp, Ld, Lt))
        wd.chunk(st, M, grad_d, act_d, x, g_dx, enc2=t_emb)                      # _time.0 = [gamma(x) | gamma(t)], _time.5 = [gamma(x) | h4], _time_out
        g = wc.finish(st) + wd.finish(st)
        return (None, None, None) + tuple(gi.to(p.dtype) for gi, p in zip(g, params))


def _tag_no_backward(out, module):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        return _NoBackward.apply(out, *list(module.parameters()))
    return out


class _PackedMixin:
    """Caches the MFMA-ordered weight stream; repacks when any parameter changed in place
    (optimizer step, load_state_dict) or moved."""

    def _init_pack(self):
        self._pack_key = None
        self._packed = None
        self._pack_bwd = {}

    def packed_bwd(self, bwd_kind=0):
        """A transposed weight stream of the backward dX chains (include/swnerf.h SWNERF_BWD_*), cached like packed()."""
        kind, names, Lp, Ld, Lt = self._pack_params()
        sd = dict(self.named_parameters())
        ps = [sd[n] for n in (_DEFORM_ORDER if bwd_kind == _lib.BWD_DEFORM else (names if bwd_kind == _lib.BWD_DNERF_FUSED else names[:24]))]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._pack_bwd.get(bwd_kind, (None, None))[0] != key:
            L = _lib.lib()
            ps32 = [p.detach() if (p.dtype == torch.float32 and p.is_contiguous()) else p.detach().float().contiguous() for p in ps]
            arr = (ctypes.c_void_p * len(ps32))(*[p.data_ptr() for p in ps32])
            buf = torch.empty(L.swnerf_packed_bwd_floats_kind(bwd_kind), dtype=torch.float32, device=ps[0].device)
            _lib.check(L.swnerf_pack_net_bwd_kind(bwd_kind, arr, Lp, Ld, _lib.ptr(buf), _lib.stream_of(buf)), "pack_net_bwd")
            self._pack_bwd[bwd_kind] = (key, buf)
        return self._pack_bwd[bwd_kind][1]

    def _wants_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _forward_train(self, x):
        """Differentiable forward (static nets)."""
        kind, names, Lp, Ld, Lt = self._pack_params()
        x = _lib.dev_f32(x, "x", self.input_ch + self.input_ch_views)
        lead = x.shape[:-1]
        flat = x.reshape(-1, x.shape[-1])
        if flat.shape[0] == 0:
            return torch.empty((*lead, 4), dtype=torch.float32, device=x.device)
        sd = dict(self.named_parameters())
        out = _MlpTrain.apply(self, flat, *[sd[n] for n in names[:24]])
        return out.reshape(*lead, 4)

    def _is_fused_arch(self):
        """True for the shape the register-resident kernels are built for; anything else runs layer by layer on the
        generic GEMM kernels (swnerf/generic.py)."""
        try:
            self._pack_params()
            return True
        except NotImplementedError:
            return False

    def _check_arch(self):
        if not (self.D == 8 and self.W == 256 and list(self.skips) == [4] and self.use_viewdirs):
            raise NotImplementedError(
                f"swnerf: only D=8, W=256, skips=[4], use_viewdirs=True is built as a HIP kernel "
                f"(got D={self.D}, W={self.W}, skips={self.skips}, use_viewdirs={self.use_viewdirs})")
        Lp, Ld = _bands(self.input_ch, 3), _bands(self.input_ch_views, 3)
        if Lp is None or Ld is None or Lp > 10 or Ld > 4:
            raise NotImplementedError(
                f"swnerf: input_ch={self.input_ch}/input_ch_views={self.input_ch_views} must be 3*(1+2L) with "
                "L<=10 / L<=4 (the reference's get_embedder output sizes)")
        return Lp, Ld

    def _pack_params(self):
        raise NotImplementedError

    def _noview_params(self):
        """(names, L_pos, out_ch) when this is the 8x256 / skips=[4] net WITHOUT view directions (use_viewdirs=False, the
        reference's argparse default: model.py:59-60, outputs = output_linear(h) with 4 or 5 channels, nerf/run.py:231) -
        the shape the fused render pass has a variant for (SWNERF_NET_NOVIEW); else None."""
        if not (self.D == 8 and self.W == 256 and list(self.skips) == [4] and not self.use_viewdirs and hasattr(self, "output_linear")):
            return None
        Lp = _bands(self.input_ch, 3)
        out_ch = self.output_linear.out_features
        if Lp is None or Lp > 10 or out_ch not in (4, 5):
            return None
        return _NOVIEW_ORDER, Lp, out_ch

    def packed_noview(self):
        """(packed float tensor of kind SWNERF_NET_NOVIEW, L_pos, out_ch), cached like packed()."""
        names, Lp, out_ch = self._noview_params()
        sd = dict(self.named_parameters())
        ps = [sd[n] for n in names]
        if not ps[0].is_cuda:
            raise RuntimeError("swnerf: module parameters must be on the GPU (call .to('cuda')); no CPU fallback")
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._pack_key:
            L = _lib.lib()
            ps32 = [p.detach() if (p.dtype == torch.float32 and p.is_contiguous()) else p.detach().float().contiguous() for p in ps]
            arr = (ctypes.c_void_p * len(ps32))(*[p.data_ptr() for p in ps32])
            buf = torch.empty(L.swnerf_packed_floats(_lib.NET_NOVIEW), dtype=torch.float32, device=ps[0].device)
            _lib.check(L.swnerf_pack_net_noview(arr, Lp, out_ch, _lib.ptr(buf), _lib.stream_of(buf)), "pack_net_noview")
            self._packed, self._pack_key = buf, key
        return self._packed, Lp, out_ch

    def packed_bwd_noview(self):
        """The transposed stream of the NOVIEW net's dX chain (swnerf_pack_net_bwd_noview), cached like packed_bwd()."""
        names, Lp, out_ch = self._noview_params()
        sd = dict(self.named_parameters())
        ps = [sd[n] for n in names]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._pack_bwd.get("noview", (None, None))[0] != key:
            L = _lib.lib()
            ps32 = [p.detach() if (p.dtype == torch.float32 and p.is_contiguous()) else p.detach().float().contiguous() for p in ps]
            arr = (ctypes.c_void_p * len(ps32))(*[p.data_ptr() for p in ps32])
            buf = torch.empty(L.swnerf_packed_bwd_noview_floats(), dtype=torch.float32, device=ps[0].device)
            _lib.check(L.swnerf_pack_net_bwd_noview(arr, Lp, out_ch, _lib.ptr(buf), _lib.stream_of(buf)), "pack_net_bwd_noview")
            self._pack_bwd["noview"] = (key, buf)
        return self._pack_bwd["noview"][1]

    def packed(self):
        """(kind, packed float tensor, L_pos, L_dir, L_time)"""
        kind, names, Lp, Ld, Lt = self._pack_params()
        sd = dict(self.named_parameters())
        ps = [sd[n] for n in names]
        dev = ps[0].device
        if not ps[0].is_cuda:
            raise RuntimeError("swnerf: module parameters must be on the GPU (call .to('cuda')); no CPU fallback")
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._pack_key:
            L = _lib.lib()
            ps32 = [p.detach() if (p.dtype == torch.float32 and p.is_contiguous()) else p.detach().float().contiguous() for p in ps]
            arr = (ctypes.c_void_p * len(ps32))(*[p.data_ptr() for p in ps32])
            buf = torch.empty(L.swnerf_packed_floats(kind), dtype=torch.float32, device=dev)
            _lib.check(L.swnerf_pack_net(kind, arr, Lp, Ld, Lt, _lib.ptr(buf), _lib.stream_of(buf)), "pack_net")
            self._packed, self._pack_key = buf, key
        return kind, self._packed, Lp, Ld, Lt

    def packed_x3(self):
        """The bf16x3 weight stream of this net (include/swnerf.h swnerf_pack_net_x3_kind), cached like packed()."""
        kind, packed, Lp, Ld, Lt = self.packed()
        if getattr(self, "_pack_x3_key", None) != self._pack_key:
            L = _lib.lib()
            _, names, _, _, _ = self._pack_params()
            sd = dict(self.named_parameters())
            ps32 = [p.detach() if (p.dtype == torch.float32 and p.is_contiguous()) else p.detach().float().contiguous()
                    for p in (sd[n] for n in names)]
            arr = (ctypes.c_void_p * len(ps32))(*[p.data_ptr() for p in ps32])
            buf = torch.empty(L.swnerf_packed_x3_floats_kind(kind), dtype=torch.float32, device=packed.device)
            _lib.check(L.swnerf_pack_net_x3_kind(kind, arr, Lp, Ld, Lt, _lib.ptr(packed), _lib.ptr(buf), _lib.stream_of(buf)), "pack_net_x3")
            self._pack_x3, self._pack_x3_key = buf, self._pack_key
        return self._pack_x3, Lp, Ld

    def _forward_hip(self, x, t_emb=None, run_deform=0, want_dx=False):
        kind, packed, Lp, Ld, Lt = self.packed()
        x = _lib.dev_f32(x, "x", self.input_ch + self.input_ch_views)
        lead = x.shape[:-1]
        flat = x.reshape(-1, x.shape[-1])
        M = flat.shape[0]
        out = torch.empty((M, 4), dtype=torch.float32, device=x.device)
        dx = torch.empty((M, 3), dtype=torch.float32, device=x.device) if want_dx else None
        if t_emb is not None:
            t_emb = _lib.dev_f32(t_emb, "ts[0]", 1 + 2 * Lt).reshape(-1, 1 + 2 * Lt)
        _lib.check(_lib.lib().swnerf_mlp_forward(kind, _lib.ptr(packed), _lib.ptr(flat), M, Lp, Ld, _lib.ptr(t_emb), Lt,
                                                 int(run_deform), _lib.ptr(out), _lib.ptr(dx), _lib.stream_of(x)),
                   "mlp_forward")
        out = _tag_no_backward(out.reshape(*lead, 4), self)
        return out, (dx.reshape(*lead, 3) if want_dx else None)


def _build_layers(mod, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, output_color_ch=3):
    mod.pts_linears = nn.ModuleList(
        [nn.Linear(input_ch, W)] + [nn.Linear(W, W) if i not in skips else nn.Linear(W + input_ch, W) for i in range(D - 1)])
    mod.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
    if use_viewdirs:
        mod.feature_linear = nn.Linear(W, W)
        mod.alpha_linear = nn.Linear(W, 1)
        mod.rgb_linear = nn.Linear(W // 2, output_color_ch)
    else:
        mod.output_linear = nn.Linear(W, output_ch)


class vallina_NeRF(nn.Module, _PackedMixin):
    """model.py:10-62."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.skips, self.use_viewdirs = skips, use_viewdirs
        _build_layers(self, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs)
        self._init_pack()

    def _pack_params(self):
        Lp, Ld = self._check_arch()
        return _lib.NET_CANON, _CANON_ORDER, Lp, Ld, 0

    def forward(self, x):
        if not self._is_fused_arch():
            if self._noview_params() is not None and not self._wants_grad() and isinstance(x, torch.Tensor) and x.is_cuda:
                return self._forward_noview(x)           # use_viewdirs=False at 8x256: the register-resident trunk + output_linear heads
            from .generic import canonical_forward
            return canonical_forward(self, x)
        if self._wants_grad():
            return self._forward_train(x)
        return self._forward_hip(x)[0]

    def _forward_noview(self, x):
        packed, Lp, out_ch = self.packed_noview()
        x = _lib.dev_f32(x, "x", self.input_ch + self.input_ch_views)
        lead = x.shape[:-1]
        flat = x.reshape(-1, x.shape[-1])
        out = torch.empty((flat.shape[0], out_ch), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().swnerf_mlp_forward_noview(_lib.ptr(packed), _lib.ptr(flat), flat.shape[0], flat.shape[1], Lp, out_ch,
                                                        _lib.ptr(out), _lib.stream_of(x)), "mlp_forward_noview")
        return out.reshape(*lead, out_ch)


class NeRFOriginal(nn.Module, _PackedMixin):
    """model.py:227-296: same network, kaiming-normal weights, returns (out, zeros[M,3])."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, input_ch_time=1, output_ch=4, skips=[4],
                 use_viewdirs=False, memory=[], embed_fn=None, output_color_ch=3, zero_canonical=True):
        super().__init__()
        if any(i in memory for i in range(D - 1)):
            raise NotImplementedError                                   # model.py:243-244
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.skips, self.use_viewdirs = skips, use_viewdirs
        _build_layers(self, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, output_color_ch)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in')  # model.py:270-272
        self._init_pack()

    def _pack_params(self):
        Lp, Ld = self._check_arch()
        return _lib.NET_CANON, _CANON_ORDER, Lp, Ld, 0

    def forward(self, x, ts):
        if not self._is_fused_arch():
            from .generic import canonical_forward
            return canonical_forward(self, x), torch.zeros_like(x[..., :3])
        out = self._forward_train(x) if self._wants_grad() else self._forward_hip(x)[0]
        return out, torch.zeros_like(x[..., :3])


class DirectTemporalNeRF(nn.Module, _PackedMixin):
    """model.py:93-151: deformation net `_time`/`_time_out`, then the canonical `_occ`."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, input_ch_time=1, output_ch=4, skips=[4],
                 use_viewdirs=False, memory=[], embed_fn=None, zero_canonical=True):
        super().__init__()
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.input_ch_time, self.skips, self.use_viewdirs = input_ch_time, skips, use_viewdirs
        self.memory, self.embed_fn, self.zero_canonical = memory, embed_fn, zero_canonical
        self._occ = NeRFOriginal(D=D, W=W, input_ch=input_ch, input_ch_views=input_ch_views,
                                 input_ch_time=input_ch_time, output_ch=output_ch, skips=skips,
                                 use_viewdirs=use_viewdirs, memory=memory, embed_fn=embed_fn, output_color_ch=3)
        layers = [nn.Linear(input_ch + input_ch_time, W)]
        for i in range(D - 1):
            if i in memory:
                raise NotImplementedError
            layers.append(nn.Linear(W + input_ch if i in skips else W, W))
        self._time = nn.ModuleList(layers)
        self._time_out = nn.Linear(W, 3)
        self._init_pack()

    def _pack_params(self):
        Lp, Ld = self._check_arch()
        Lt = _bands(self.input_ch_time, 1)
        if Lt is None or Lt > 10:
            raise NotImplementedError(f"swnerf: input_ch_time={self.input_ch_time} must be 1+2L with L<=10")
        emb = self.embed_fn
        if emb is not None and getattr(emb, "multires", Lp) != Lp:
            raise NotImplementedError("swnerf: DirectTemporalNeRF.embed_fn must be the get_embedder(multires, 3) "
                                      "encoder matching input_ch (the kernel re-embeds x+dx itself, model.py:148-149)")
        return _lib.NET_DNERF, ["_occ." + n for n in _CANON_ORDER] + _DEFORM_ORDER, Lp, Ld, Lt

    def forward(self, x, ts):
        if not self._is_fused_arch():
            from .generic import temporal_forward
            return temporal_forward(self, x, ts)
        t = ts[0]
        # the reference asserts one unique time and branches on its value: two host syncs
        # (model.py:141-144); one here
        lo, hi = torch.aminmax(t[:, :1])
        lo, hi = float(lo), float(hi)
        assert lo == hi, "Only accepts all points from same time"
        run_deform = not (lo == 0. and self.zero_canonical)
        if self._wants_grad():
            if not run_deform:                                                  # model.py:143-145: canonical net only, dx = 0
                out, dx = self._occ(x, ts)
                return out, dx
            kind, names, Lp, Ld, Lt = self._pack_params()
            x = _lib.dev_f32(x, "x", self.input_ch + self.input_ch_views)
            lead = x.shape[:-1]
            flat = x.reshape(-1, x.shape[-1])
            te = _lib.dev_f32(t, "ts[0]", 1 + 2 * Lt).reshape(-1, 1 + 2 * Lt)
            if flat.shape[0] == 0:
                return (torch.empty((*lead, 4), dtype=torch.float32, device=x.device),
                        torch.empty((*lead, 3), dtype=torch.float32, device=x.device))
            sd = dict(self.named_parameters())
            out, dx = _DnerfTrain.apply(self, flat, te, *[sd[n] for n in names])
            return out.reshape(*lead, 4), dx.reshape(*lead, 3)
        out, dx = self._forward_hip(x, t_emb=t, run_deform=run_deform, want_dx=True)
        return out, dx


class NeRF:
    @staticmethod
    def get_by_name(type, *args, **kwargs):
        """model.py:214-225."""
        print("NeRF type selected: %s" % type)
        if type == "original":
            return NeRFOriginal(*args, **kwargs)
        if type == "direct_temporal":
            return DirectTemporalNeRF(*args, **kwargs)
        raise ValueError("Type %s not recognized." % type)
