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
    @staticmethod
    def forward(ctx, out, *params):
        return out.view_as(out)

    @staticmethod
    def backward(ctx, *g):
        raise NotImplementedError(
            "swnerf: the backward pass of the fused NeRF MLP is not built yet (SURVEY.md 8f rank 1); "
            "run the render path under torch.no_grad()")


def _zero_grads(params):
    """fp32 gradient buffers for `params`, views of ONE zeroed allocation (one fill instead of one per tensor).  Every
    view starts 16-byte aligned, as the GEMM kernels' vector paths want."""
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
    return [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)]


GEMM_STREAMS = int(os.environ.get("SWNERF_GEMM_STREAMS", "2"))    # side streams the weight-gradient GEMMs of a chunk fan out over (0/1: off)
_SIDE_STREAMS = {}


class _Fan:
    """The weight-gradient GEMMs of one row chunk are independent of each other (each reads grad / act and accumulates into
    its own C with atomics), but every launch ends with an epilogue of 64 K float atomics per workgroup (~50 us chip-wide)
    during which the matrix pipe idles, and starts with a ramp.  Issued round-robin on a few side streams, a GEMM's
    workgroups start on the CUs the previous GEMM's workgroups have left (each needs a whole CU: >128 KB of LDS), so one
    launch's epilogue runs under the next one's main loop.  fork(): the side streams wait for the current stream (the
    backward kernel that produced `grad`); next(): the stream handle for the next launch; join(): the current stream waits
    for all of them (before `grad` is overwritten / the gradients are read)."""

    def __init__(self, device):
        self.main = torch.cuda.current_stream(device)
        n = GEMM_STREAMS if GEMM_STREAMS > 1 else 0
        key = (device.index, n)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
        self.side = _SIDE_STREAMS[key]
        self.i = 0

    def fork(self):
        if self.side:
            ev = torch.cuda.Event()
            ev.record(self.main)
            for s in self.side:
                s.wait_event(ev)

    def next(self):
        if not self.side:
            return ctypes.c_void_p(self.main.cuda_stream)
        s = self.side[self.i % len(self.side)]
        self.i += 1
        return ctypes.c_void_p(s.cuda_stream)

    def join(self):
        for s in self.side:
            ev = torch.cuda.Event()
            ev.record(s)
            self.main.wait_event(ev)


def _st(st):
    return st.next() if isinstance(st, _Fan) else st


def _gemm_tn(L, st, M, A, a_col, No, B, b_col, Ni, C, c_col, bias):
    """C[:, c_col:c_col+Ni] += A[:, a_col:a_col+No]^T . B[:, b_col:b_col+Ni];  bias += column sums of that A block"""
    if isinstance(st, _Group):
        if No == 256 and Ni == 256:
            return _gemm_tn_fused(L, st, M, A, a_col, B, b_col, C, c_col, bias)
        st = st.st                                            # narrow shapes keep their own launches
    _lib.check(L.swnerf_gemm_tn(A.data_ptr() + 4 * a_col, A.stride(0), No, B.data_ptr() + 4 * b_col, B.stride(0), Ni, M,
                                C.data_ptr() + 4 * c_col, C.stride(0), _lib.ptr(bias), _st(st)), "gemm_tn")


class _Group:
    """Collects the 256 x 256 weight-gradient GEMMs of one row chunk and launches them as ONE kernel (swnerf_gemm_tn_group:
    one ramp and one atomic epilogue per chunk instead of one per layer); `st` is where that launch goes."""

    def __init__(self, st):
        self.st, self.items = st, []

    def launch(self, L, M):
        if self.items:
            arr = (_lib.GemmItem * len(self.items))(*self.items)
            _lib.check(L.swnerf_gemm_tn_group(arr, len(self.items), M, _st(self.st)), "gemm_tn_group")
            self.items = []


# SWNERF_GEMM_GROUP: 1 (default) = the 256 x 256 GEMMs of a chunk share a launch (_chunk_gemms); plain = the skip layer's GEMM
# (gamma(x) rider: its workgroups run 1.2-1.3x longer) keeps its own launch; 0 = one launch per layer (profiles/r03/gemm_group.md)
GEMM_GROUP = os.environ.get("SWNERF_GEMM_GROUP", "1") != "0"
NARROW_FUSED = os.environ.get("SWNERF_NARROW_FUSED", "1") != "0"     # the canonical net's five narrow weight-gradient products as one kernel
GROUP_RIDERS = os.environ.get("SWNERF_GEMM_GROUP", "1") != "plain"  # the skip layer's GEMM (gamma(x) rider) joins the group, at work weight 6 : 4      # 0: one launch per layer (round 2 / early round 3)


def _gemm_tn_fused(L, st, M, A, a_col, B, b_col, C, c_col, bias, B2=None, b2_col=0, Ni2=0, C2=None, c2_col=0,
                   A2=None, a2_col=0, No2=0, C3=None, bias3=None):
    """256x256 block C[:, c_col:] += A[:, a_col:]^T . B[:, b_col:] with the riders of swnerf_gemm_tn_fused.
    st a _Group: queued for the chunk's grouped launch."""
    off = lambda T_, col: None if T_ is None else T_.data_ptr() + 4 * col
    ld = lambda T_: 0 if T_ is None else T_.stride(0)
    if isinstance(st, _Group):
        st.items.append(_lib.GemmItem(off(A, a_col), ld(A), off(B, b_col), ld(B), off(C, c_col), ld(C), _lib.ptr(bias),
                                      off(B2, b2_col), ld(B2), Ni2, off(C2, c2_col), ld(C2),
                                      off(A2, a2_col), ld(A2), No2, off(C3, 0), ld(C3), _lib.ptr(bias3)))
        return
    _lib.check(L.swnerf_gemm_tn_fused(off(A, a_col), ld(A), off(B, b_col), ld(B), M, off(C, c_col), ld(C), _lib.ptr(bias),
                                      off(B2, b2_col), ld(B2), Ni2, off(C2, c2_col), ld(C2),
                                      off(A2, a2_col), ld(A2), No2, off(C3, 0), ld(C3), _lib.ptr(bias3), _st(st)), "gemm_tn_fused")


def _rgb4_buffers(device):
    """Zeroed 4-row accumulators (weight [4,128], bias [4]) for rgb_linear's gradient, see _rgb_weight_grad."""
    return (torch.zeros((4, 128), dtype=torch.float32, device=device), torch.zeros((4,), dtype=torch.float32, device=device))


def _rgb_weight_grad(L, st, M, d_out, act, g, rgb4):
    """rgb_linear: dW [3,128] += d rgb^T . hv.  d_out is [M,4] = [d rgb(3), d sigma]; taking all FOUR columns as output
    rows keeps the operand 16-byte aligned with a column count that is a multiple of 4, i.e. on the LDS-DMA staged
    variant of the GEMM kernel (the 3-column form falls to 4-byte loads): it accumulates into the 4-row buffers
    `rgb4`, whose first three rows the caller hands out as the gradient (_rgb4_finish); the 4th row is dropped."""
    if rgb4 is None or d_out.stride(0) != 4 or d_out.data_ptr() % 16:
        _gemm_tn(L, st, M, d_out, 0, 3, act, SW_ACT_HV, 128, g[22], 0, g[23])
    else:
        _gemm_tn(L, st, M, d_out, 0, 4, act, SW_ACT_HV, 128, rgb4[0], 0, rgb4[1])


def _rgb4_finish(g, rgb4):
    g[22] = g[22] + rgb4[0][:3]
    g[23] = g[23] + rgb4[1][:3]


SW_ACT_HV = 2304          # csrc/swnerf_common.h: column of the view hidden layer in the act / grad rows


def _feature_buffers(device):
    """Zeroed accumulators around feature_linear for the op path: G = d pre_hv^T . h7 [128, 256] and the 4-row form of
    alpha_linear's gradient (A = d raw [M, 4]; row 3) with its bias - see _feature_finish."""
    z = torch.zeros(128 * 256 + 4 * 256 + 4, dtype=torch.float32, device=device)
    return z[:32768].view(128, 256), z[32768:33792].view(4, 256), z[33792:33796]


def _canon_weight_grads(L, st, M, grad, act, x, d_out, Cpos, Cdir, g, rgb4, fbufs):
    """dW / db of the 12 Linear layers of the canonical net (g: zeroed fp32 tensors in _CANON_ORDER) from the
    dX chain's `grad`, the saved `act`, the embedded inputs x = [gamma(x) | gamma(d)] and d raw.  feature_linear is folded
    into the view layer in every kernel (csrc/swnerf_common.h SW_CANON_STEPS), so neither `feature` nor d feature exists:
    its two neighbours' gradients come from G (fbufs; finished by _feature_finish)."""
    gfeat, a4w, a4b = fbufs
    mm = lambda A, a_col, No, B, b_col, Ni, wi, c_col, with_bias: _gemm_tn(
        L, st, M, A, a_col, No, B, b_col, Ni, g[wi], c_col, g[wi + 1] if with_bias else None)
    mm(grad, 0, 256, x, 0, Cpos, 0, 0, True)                                   # pts_linears.0
    for l in (1, 2, 3, 4, 6, 7):
        mm(grad, 256 * l, 256, act, 256 * (l - 1), 256, 2 * l, 0, True)
    # pts_linears.5 = [pts | h4]: one pass over d pre_5 for both column blocks
    _gemm_tn_fused(L, st, M, grad, 1280, act, 1024, g[10], Cpos, g[11], B2=x, b2_col=0, Ni2=Cpos, C2=g[10], c2_col=0)
    _gemm_tn(L, st, M, grad, SW_ACT_HV, 128, act, 1792, 256, gfeat, 0, g[17])  # G (+ views_linears.0.bias)
    mm(grad, SW_ACT_HV, 128, x, Cpos, Cdir, 16, 256, False)                    # views_linears.0, gamma(d) columns
    _gemm_tn(L, st, M, d_out, 0, 4, act, 1792, 256, a4w, 0, a4b)               # alpha_linear = row 3 of d raw^T . h7
    _rgb_weight_grad(L, st, M, d_out, act, g, rgb4)


def _feature_finish(L, st, gfeat, a4w, a4b, g, params):
    """Everything that hangs on feature_linear, from G = sum_rows d pre_hv (x) h7 (params / g in _CANON_ORDER):
      feature = W_f h7 + b_f   =>  d views_linears.0.weight[:, :256] = sum d pre_hv (x) feature = G W_f^T + db_hv (x) b_f
      d feature = Wv_f^T d pre_hv  =>  d feature_linear.weight = sum d feature (x) h7 = Wv_f^T G,  d feature_linear.bias = Wv_f^T db_hv
    (Wv_f = views_linears.0.weight[:, :256]; three 128 x 256 x 256 products per step instead of 2 KB of stores, 2 KB of loads
    and 131 kFLOP per row), and alpha_linear from row 3 of the 4-row form.  model.py:49-53."""
    f32 = lambda p_: p_.detach() if (p_.dtype == torch.float32 and p_.is_contiguous()) else p_.detach().float().contiguous()
    Wv, W_f, b_f = f32(params[16]), f32(params[18]), f32(params[19])
    _lib.check(L.swnerf_feature_finish(_lib.ptr(gfeat), _lib.ptr(g[17]), _lib.ptr(Wv), Wv.stride(0), _lib.ptr(W_f), _lib.ptr(b_f), _lib.ptr(a4w),
                                       _lib.ptr(a4b), _lib.ptr(g[16]), g[16].stride(0), _lib.ptr(g[18]), _lib.ptr(g[19]), _lib.ptr(g[20]),
                                       _lib.ptr(g[21]), st), "feature_finish")


def _slot_buffers(device):
    """Zeroed accumulators of the fused training pass: slot-ordered columns for the three encoding GEMMs, G = d pre_hv^T . h7
    [128, 256] (from which BOTH feature-related weight gradients follow, _unslot_weight_grads) and the 4-row form of
    alpha_linear's gradient (A = d raw [rows, 4]; row 3) with its bias."""
    z = torch.zeros(256 * 64 + 256 * 64 + 128 * 32 + 128 * 256 + 4 * 256 + 4, dtype=torch.float32, device=device)
    o = [0, 16384, 32768, 36864, 69632, 70656, 70660]
    return (z[o[0]:o[1]].view(256, 64), z[o[1]:o[2]].view(256, 64), z[o[2]:o[3]].view(128, 32),
            z[o[3]:o[4]].view(128, 256), z[o[4]:o[5]].view(4, 256), z[o[5]:o[6]])


def _trunk_plain_grads(L, st, M, grad, act, g):
    """The six rider-free 256 x 256 weight gradients of an 8 x 256 trunk (layers 1-4, 6, 7; model.py:39-47 reversed)."""
    for l in (1, 2, 3, 4, 6, 7):
        _gemm_tn(L, st, M, grad, 256 * l, 256, act, 256 * (l - 1), 256, g[2 * l], 0, g[2 * l + 1])


def _canon_weight_grads_slots(L, st, M, grad, act, xs, d_out, Cpos, Cdir, g, slot_bufs, rgb4=None, part="all"):
    """The same 12 weight gradients for the FUSED training pass (accumulating: call once per row chunk): the
    encodings come as xs [M, 96] in operand slot order (64 slots gamma(x), 32 slots gamma(d); csrc/swnerf_common.h
    sw_xs_col), so the three GEMMs against them accumulate slot-ordered columns into `slot_bufs`, which
    _unslot_weight_grads moves to their reference columns at the end.  Every operand is 16-byte aligned here
    (x[:, :63] with ld 90 was not).  part: "plain" = the six rider-free 256 x 256 GEMMs only (for the chunk's grouped launch,
    _chunk_gemms), "rest" = everything else, "all" = both."""
    c0s, c5s, cvs, gfeat, a4w, a4b = slot_bufs
    l5 = lambda: _gemm_tn_fused(L, st, M, grad, 1280, act, 1024, g[10], Cpos, g[11], B2=xs, b2_col=0, Ni2=64, C2=c5s, c2_col=0)   # pts_linears.5
    if part != "rest":
        _trunk_plain_grads(L, st, M, grad, act, g)
        if part == "all" or GROUP_RIDERS:
            l5()
    if part == "plain":
        return
    mm = lambda A, a_col, No, B, b_col, Ni, C, c_col, bias: _gemm_tn(L, st, M, A, a_col, No, B, b_col, Ni, C, c_col, bias)
    if part == "rest" and not GROUP_RIDERS:
        l5()
    if (NARROW_FUSED and rgb4 is not None and d_out.stride(0) == 4 and grad.stride(0) == act.stride(0) and xs.stride(0) == 96
            and not (grad.data_ptr() | act.data_ptr() | xs.data_ptr() | d_out.data_ptr()) % 16):
        # the five narrow products below as ONE pass over the rows (csrc/backward_kernels.hip narrow5_kernel)
        _lib.check(L.swnerf_canon_narrow_grads(_lib.ptr(grad), grad.stride(0), _lib.ptr(act), act.stride(0), _lib.ptr(xs), _lib.ptr(d_out), M,
                                               _lib.ptr(c0s), _lib.ptr(cvs), _lib.ptr(gfeat), _lib.ptr(a4w), _lib.ptr(rgb4[0]), _lib.ptr(g[1]),
                                               _lib.ptr(g[17]), _lib.ptr(a4b), _lib.ptr(rgb4[1]), _st(st)), "canon_narrow_grads")
        return
    mm(grad, 0, 256, xs, 0, 64, c0s, 0, g[1])                                  # pts_linears.0
    # feature_linear has NO activation (model.py:50-51: feature = feature_linear(h); h = cat[feature, views]), so both weight
    # gradients around it are linear images of ONE small matrix, G = d pre_hv^T . h7 [128, 256] (_unslot_weight_grads):
    # neither `feature` nor d feature is ever stored or read, and the 256 x 256 GEMM of feature_linear is not run at all.
    mm(grad, 2304, 128, act, 1792, 256, gfeat, 0, g[17])                       # G (+ views_linears.0.bias)
    mm(grad, 2304, 128, xs, 64, 32, cvs, 0, None)                              # views_linears.0, gamma(d) slots
    mm(d_out, 0, 4, act, 1792, 256, a4w, 0, a4b)                               # alpha_linear = row 3 of d raw^T . h7
    _rgb_weight_grad(L, st, M, d_out, act, g, rgb4)


def _chunk_gemms(L, fan, M, jobs, rest_on_main=False):
    """The weight-gradient GEMMs of one row chunk.  jobs: callables job(st, part).  With SWNERF_GEMM_GROUP (default) the
    rider-free 256 x 256 GEMMs of all jobs go out first as ONE launch on the main stream, alone on the chip (its workgroups
    run ~2.5 ms each: next to another kernel they would start in rounds and finish in rounds, with half the chip idle in
    between - measured +2.8 ms on the step without view directions), then the rest fans out over the side streams."""
    if GEMM_GROUP:
        grp = _Group(ctypes.c_void_p(fan.main.cuda_stream))
        for job in jobs:
            job(grp, "plain")
        grp.launch(L, M)
        # a job whose rest is ONE launch (swnerf_canon_narrow_grads) keeps it on the main stream, behind the group; what is
        # left fans out over the side streams
        on_main = rest_on_main if isinstance(rest_on_main, (list, tuple)) else [rest_on_main] * len(jobs)
        for job, m_ in zip(jobs, on_main):
            if m_:
                job(grp.st, "rest")
        if all(on_main):
            return
        fan.fork()
        for job, m_ in zip(jobs, on_main):
            if not m_:
                job(fan, "rest")
    else:
        fan.fork()
        for job in jobs:
            job(fan, "all")
    fan.join()                                               # before the next chunk's backward kernel overwrites grad / d_raw


def _noview_slot_buffers(device):
    """Zeroed accumulators of the fused NOVIEW training pass: slot-ordered gamma(x) columns of pts_linears.0 / .5 and the
    8-row form of output_linear's gradient (A = d raw [rows, 8]) with its bias."""
    z = torch.zeros(256 * 64 + 256 * 64 + 8 * 256 + 8, dtype=torch.float32, device=device)
    return z[:16384].view(256, 64), z[16384:32768].view(256, 64), z[32768:34816].view(8, 256), z[34816:34824]


def _noview_weight_grads_slots(L, st, M, grad, act, xs, d_raw8, Cpos, g, bufs, part="all"):
    """dW / db of the 8x256 net without view directions (g: zeroed tensors in _NOVIEW_ORDER), accumulating per row chunk:
    model.py:39-47,59-60 reversed.  grad / act columns 0..2047 = pts_linears.0..7; xs slots 0..63 = gamma(x).  part as above."""
    c0s, c5s, w8, b8 = bufs
    l5 = lambda: _gemm_tn_fused(L, st, M, grad, 1280, act, 1024, g[10], Cpos, g[11], B2=xs, b2_col=0, Ni2=64, C2=c5s, c2_col=0)   # pts_linears.5
    if part != "rest":
        _trunk_plain_grads(L, st, M, grad, act, g)
        if part == "all" or GROUP_RIDERS:
            l5()
    if part == "plain":
        return
    if part == "rest" and not GROUP_RIDERS:
        l5()
    _gemm_tn(L, st, M, grad, 0, 256, xs, 0, 64, c0s, 0, g[1])                  # pts_linears.0
    _gemm_tn(L, st, M, d_raw8, 0, 8, act, 1792, 256, w8, 0, b8)                # output_linear (rows 0..out_ch-1)


def _noview_unslot(L, st, bufs, Lp, g):
    c0s, c5s, w8, b8 = bufs
    for cs, W in ((c0s, g[0]), (c5s, g[10])):
        _lib.check(L.swnerf_unslot_grad(_lib.ptr(cs), 64, 256, 0, 64, Lp, 0, W.data_ptr(), W.stride(0), 0, st), "unslot_grad")
    oc = g[16].shape[0]
    g[16] = g[16] + w8[:oc]
    g[17] = g[17] + b8[:oc]


def _unslot_weight_grads(L, st, slot_bufs, Lp, Ld, g, params):
    """Finish the canonical net's gradients of a fused training pass (params: its tensors in _CANON_ORDER): slot-ordered
    columns to their reference columns, then everything that hangs on feature_linear (_feature_finish)."""
    c0s, c5s, cvs, gfeat, a4w, a4b = slot_bufs
    for cs, nslots, slot0, W, col0 in ((c0s, 64, 0, g[0], 0), (c5s, 64, 0, g[10], 0), (cvs, 32, 64, g[16], 256)):
        _lib.check(L.swnerf_unslot_grad(_lib.ptr(cs), cs.stride(0), cs.shape[0], slot0, nslots, Lp, Ld, W.data_ptr(), W.stride(0),
                                        col0, st), "unslot_grad")
    _feature_finish(L, st, gfeat, a4w, a4b, g, params)


def _deform_slot_buffers(device):
    """Zeroed slot-ordered accumulators for the deformation net's encoding GEMMs (gamma(x) of `_time.0` and `_time.5`,
    gamma(t) of `_time.0`) and the 4-row form of `_time_out`'s gradient."""
    z = torch.zeros(256 * 64 + 256 * 64 + 256 * 32 + 4 * 256 + 4, dtype=torch.float32, device=device)
    return (z[:16384].view(256, 64), z[16384:32768].view(256, 64), z[32768:40960].view(256, 32),
            z[40960:41984].view(4, 256), z[41984:41988])


def _deform_weight_grads_slots(L, st, M, grad_d, act_d, xs_d, g_dx, Cpos, gd, bufs, part="all"):
    """dW / db of the deformation net (`_time.0..7`, `_time_out`; gd: zeroed tensors in _DEFORM_ORDER) for the fused D-NeRF
    training pass, accumulating (one call per row chunk): xs_d [M, 96] = gamma(x) (64 slots) and gamma(t) (32 slots) in
    operand slot order, g_dx [M, 4] = d dx with a zero 4th column (aligned: the 4-row form, 4th row dropped).  part as above."""
    c0s, c5s, cts, w4, b4 = bufs
    l5 = lambda: _gemm_tn_fused(L, st, M, grad_d, 1280, act_d, 1024, gd[10], Cpos, gd[11], B2=xs_d, b2_col=0, Ni2=64, C2=c5s, c2_col=0)   # _time.5
    if part != "rest":
        _trunk_plain_grads(L, st, M, grad_d, act_d, gd)
        if part == "all" or GROUP_RIDERS:
            l5()
    if part == "plain":
        return
    mm = lambda A, a_col, No, B, b_col, Ni, C, c_col, bias: _gemm_tn(L, st, M, A, a_col, No, B, b_col, Ni, C, c_col, bias)
    if part == "rest" and not GROUP_RIDERS:
        l5()
    mm(grad_d, 0, 256, xs_d, 0, 64, c0s, 0, gd[1])                             # _time.0 = [gamma(x) | gamma(t)]
    mm(grad_d, 0, 256, xs_d, 64, 32, cts, 0, None)
    mm(g_dx, 0, 4, act_d, 1792, 256, w4, 0, b4)                                # _time_out (rows 0..2)


def _deform_unslot(L, st, bufs, Lp, Lt, Cpos, gd):
    c0s, c5s, cts, w4, b4 = bufs
    for cs, W in ((c0s, gd[0]), (c5s, gd[10])):
        _lib.check(L.swnerf_unslot_grad(_lib.ptr(cs), 64, 256, 0, 64, Lp, 0, W.data_ptr(), W.stride(0), 0, st), "unslot_grad")
    _lib.check(L.swnerf_unslot_grad_time(_lib.ptr(cts), 32, 256, 32, Lt, gd[0].data_ptr(), gd[0].stride(0), Cpos, st), "unslot_grad_time")
    gd[16] = gd[16] + w4[:3]
    gd[17] = gd[17] + b4[:3]


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
        g = _zero_grads(params)                                               # order: _CANON_ORDER
        rgb4, fbufs = _rgb4_buffers(x.device), _feature_buffers(x.device)
        _canon_weight_grads(L, st, M, grad, act, x, d_out, module.input_ch, module.input_ch_views, g, rgb4, fbufs)
        _rgb4_finish(g, rgb4)
        _feature_finish(L, st, *fbufs, g, params)
        return (None, None) + tuple(gi.to(p.dtype) for gi, p in zip(g, params))


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
        g = _zero_grads(params)
        rgb4, fbufs = _rgb4_buffers(x.device), _feature_buffers(x.device)
        _canon_weight_grads(L, st, M, grad_c, act_c, x2, d_out, Cpos, Cdir, g, rgb4, fbufs)      # (fills g[0..23])
        _rgb4_finish(g, rgb4)
        _feature_finish(L, st, *fbufs, g, params)
        g_dx = d_pts if d_dx is None else (d_pts + d_dx.float()).contiguous()
        grad_d = torch.empty_like(act_d)
        _lib.check(L.swnerf_deform_backward_dx(_lib.ptr(module.packed_bwd(_lib.BWD_DEFORM)), _lib.ptr(bits_d), _lib.ptr(g_dx), M,
                                               _lib.ptr(grad_d), st), "deform_backward_dx")
        gd = g[24:]
        mm = lambda A, a_col, No, B, b_col, Ni, wi, c_col, with_bias: _gemm_tn(
            L, st, M, A, a_col, No, B, b_col, Ni, gd[wi], c_col, gd[wi + 1] if with_bias else None)
        mm(grad_d, 0, 256, x, 0, Cpos, 0, 0, True)                               # _time.0 = [gamma(x) | gamma(t)]
        mm(grad_d, 0, 256, t_emb, 0, Ct, 0, Cpos, False)
        for l in (1, 2, 3, 4, 6, 7):
            mm(grad_d, 256 * l, 256, act_d, 256 * (l - 1), 256, 2 * l, 0, True)
        _gemm_tn_fused(L, st, M, grad_d, 1280, act_d, 1024, gd[10], Cpos, gd[11], B2=x, b2_col=0, Ni2=Cpos, C2=gd[10],
                       c2_col=0)                                                 # _time.5 = [gamma(x) | h4]
        mm(g_dx, 0, 3, act_d, 1792, 256, 16, 0, True)                            # _time_out
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
