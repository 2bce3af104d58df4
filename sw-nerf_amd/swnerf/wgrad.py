"""Weight gradients of the 8x256 nets (model.py:39-62, 128-136 reversed): dW / db of every Linear layer
from the dX chain's `grad` rows, the saved activations `act`, the encodings and d raw - TN MFMA GEMMs over the (ray, sample) rows
(csrc/backward_kernels.hip), accumulated chunk by chunk.

ONE table-driven routine (`WeightGrads`) serves the three net kinds (canonical with view directions, without, deformation)
on both paths - the fused training passes (encodings in operand SLOT order, `xs`) and the op path (embedded rows `x`): round 3
carried three hand-unrolled copies.  Everything a backward pass accumulates into - the gradients themselves, the slot-ordered
scratch, G, the 4- / 8-row head forms - lives in ONE zeroed allocation (one fill per pass instead of five), and the head
gradients are VIEWS of their 4- / 8-row forms (no add at the end).

feature_linear: it has no activation (model.py:49-53) and every kernel runs it folded into the view layer
(csrc/swnerf_common.h SW_CANON_STEPS), so `feature` and d feature do not exist.  With G = sum_rows d pre_hv (x) h7 [128, 256]:
  d views_linears.0.weight[:, :256] = G W_f^T + db_hv (x) b_f,   d feature_linear.weight = Wv_f^T G,   d feature_linear.bias = Wv_f^T db_hv
(swnerf_feature_finish; Wv_f = views_linears.0.weight[:, :256])."""
import ctypes
import os

import torch

from . import _lib

SW_ACT_HV = 2304          # csrc/swnerf_common.h: column of the view hidden layer in the act / grad rows
SW_ACT_H7 = 1792          # ... of h7

GEMM_STREAMS = int(os.environ.get("SWNERF_GEMM_STREAMS", "2"))    # side streams the weight-gradient GEMMs of a chunk fan out over (0/1: off)
# SWNERF_GEMM_GROUP: 1 (default) = the 256 x 256 GEMMs of a chunk share a launch (_chunk_gemms); plain = the skip layer's GEMM
# (gamma(x) rider: its workgroups run 1.2-1.3x longer) keeps its own launch; 0 = one launch per layer (profiles/r03/gemm_group.md)
GEMM_GROUP = os.environ.get("SWNERF_GEMM_GROUP", "1") != "0"
GROUP_RIDERS = os.environ.get("SWNERF_GEMM_GROUP", "1") != "plain"  # the skip layer's GEMM (gamma(x) rider) joins the group, at work weight 6 : 4
NARROW_FUSED = os.environ.get("SWNERF_NARROW_FUSED", "1") != "0"     # a net's narrow weight-gradient products as one kernel
NOVIEW_NARROW_FUSED = os.environ.get("SWNERF_NOVIEW_NARROW_FUSED", "0") == "1"
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


def _gemm_tn(L, st, M, A, a_col, No, B, b_col, Ni, C, c_col, bias):
    """C[:, c_col:c_col+Ni] += A[:, a_col:a_col+No]^T . B[:, b_col:b_col+Ni];  bias += column sums of that A block"""
    if isinstance(st, _Group):
        if No == 256 and Ni == 256:
            return _gemm_tn_fused(L, st, M, A, a_col, B, b_col, C, c_col, bias)
        st = st.st                                            # narrow shapes keep their own launches
    _lib.check(L.swnerf_gemm_tn(A.data_ptr() + 4 * a_col, A.stride(0), No, B.data_ptr() + 4 * b_col, B.stride(0), Ni, M,
                                C.data_ptr() + 4 * c_col, C.stride(0), _lib.ptr(bias), _st(st)), "gemm_tn")


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
        # a job whose rest is ONE launch (the fused narrow kernels) keeps it on the main stream, behind the group; what is
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


# ---- the table: per net kind, parameter order, scratch accumulators and the narrow products ----------------------------
# A narrow product: (A source, A column, A width, B source, B column, B width, destination, destination column, bias destination)
#   sources:  "grad" d(pre-activation) rows | "act" saved activations | "enc" the encodings (xs slots or x) | "enc2" a second
#             encoding tensor (the op path's gamma(t)) | "draw" d raw / d dx rows
#   destinations: ("g", index into the parameter-ordered gradient list) or ("s", scratch name)
# Widths / columns that depend on the path (slot order vs reference columns) are filled in by WeightGrads.__init__.
KINDS = {
    # vallina_NeRF / NeRFOriginal with view directions: _CANON_ORDER (pts_linears 0..15, views 16/17, feature 18/19, alpha 20/21, rgb 22/23)
    "canon": dict(n_params=24, head=(22, 23, "rgb4w", "rgb4b", 3)),
    # vallina_NeRF with use_viewdirs=False: _NOVIEW_ORDER (pts_linears 0..15, output_linear 16/17)
    "noview": dict(n_params=18, head=(16, 17, "w8", "b8", None)),
    # the deformation net of DirectTemporalNeRF: _DEFORM_ORDER (_time 0..15, _time_out 16/17)
    "deform": dict(n_params=18, head=(16, 17, "w4", "b4", 3)),
}


class WeightGrads:
    """Accumulates dW / db of one 8x256 net over row chunks and finishes them.

        wg = WeightGrads(L, kind, params, fused=..., Cpos=..., Cdir=..., Ct=..., bands=(Lp, Ld, Lt))
        for each chunk:  wg.chunk(st, M, grad, act, enc, draw, part=..., enc2=...)      # via _chunk_gemms
        grads = wg.finish(st)            # list aligned with `params` (fp32)

    params: the net's tensors in its kind's order (zero-copy: only shapes / the current feature_linear and view weights are read).
    fused=True: `enc` is xs [M, 96] in operand slot order (gamma(x) slots 0..63, then gamma(d) or gamma(t) slots 64..95) and the
    products against it land in slot-ordered scratch that finish() moves to reference columns; fused=False (the op path):
    `enc` is the embedded row tensor x = [gamma(x) | gamma(d)] (any leading dimension) and `enc2` gamma(t) for the deformation net."""

    def __init__(self, L, kind, params, *, fused, Cpos, Cdir=0, Ct=0, bands=(0, 0, 0)):
        self.L, self.kind, self.params, self.fused = L, kind, params, fused
        self.Cpos, self.Cdir, self.Ct, self.bands = Cpos, Cdir, Ct, bands
        spec = KINDS[kind]
        assert len(params) == spec["n_params"], (kind, len(params))
        dev = params[0].device
        hw, hb, sw, sb, hrows = spec["head"]
        if hrows is None:
            hrows = params[hw].shape[0]                          # output_linear: 4 or 5 channels
        self.head = (hw, hb, sw, sb, hrows)
        scratch = {}
        if fused:
            scratch["c0s"], scratch["c5s"] = (256, 64), (256, 64)
        if kind == "canon":
            scratch.update(gfeat=(128, 256), a4w=(4, 256), a4b=(4,), rgb4w=(4, 128), rgb4b=(4,))
            if fused:
                scratch["cvs"] = (128, 32)
        elif kind == "noview":
            scratch.update(w8=(8, 256), b8=(8,))
        else:
            scratch.update(w4=(4, 256), b4=(4,))
            if fused:
                scratch["cts"] = (256, 32)
        # ONE zeroed allocation: every gradient tensor (except the head's, which are views of the 4- / 8-row forms) and every
        # scratch accumulator; each view starts 16-byte aligned, as the GEMM kernels' vector paths want
        shapes = [("g", i, tuple(p.shape)) for i, p in enumerate(params) if i not in (hw, hb)] + [("s", k, v) for k, v in scratch.items()]
        offs, n = [], 0
        for _, _, shp in shapes:
            offs.append(n)
            cnt = 1
            for d in shp:
                cnt *= d
            n += (cnt + 3) // 4 * 4
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.g, self.s = [None] * len(params), {}
        for (where, key, shp), o in zip(shapes, offs):
            cnt = 1
            for d in shp:
                cnt *= d
            v = flat[o:o + cnt].view(shp)
            if where == "g":
                self.g[key] = v
            else:
                self.s[key] = v
        self.g[hw], self.g[hb] = self.s[sw][:hrows], self.s[sb][:hrows]

    # -- one row chunk ---------------------------------------------------------------------------------------------------
    def chunk(self, st, M, grad, act, enc, draw, part="all", enc2=None):
        """part: "plain" = the six rider-free 256 x 256 GEMMs only (for the chunk's grouped launch, _chunk_gemms), "rest" =
        everything else, "all" = both."""
        L, g, s, Cpos = self.L, self.g, self.s, self.Cpos
        e0 = 64 if self.fused else Cpos                          # width of the gamma(x) block of `enc`
        c5 = (s["c5s"], 0) if self.fused else (g[10], 0)         # where the skip layer's gamma(x) columns accumulate
        l5 = lambda: _gemm_tn_fused(L, st, M, grad, 1280, act, 1024, g[10], Cpos, g[11], B2=enc, b2_col=0, Ni2=e0, C2=c5[0], c2_col=c5[1])
        if part != "rest":
            for l in (1, 2, 3, 4, 6, 7):
                _gemm_tn(L, st, M, grad, 256 * l, 256, act, 256 * (l - 1), 256, g[2 * l], 0, g[2 * l + 1])
            if part == "all" or GROUP_RIDERS:
                l5()
        if part == "plain":
            return
        if part == "rest" and not GROUP_RIDERS:
            l5()
        mm = lambda A, a_col, No, B, b_col, Ni, C, c_col, bias: _gemm_tn(L, st, M, A, a_col, No, B, b_col, Ni, C, c_col, bias)
        c0 = (s["c0s"], 0) if self.fused else (g[0], 0)
        aligned4 = draw.stride(0) == 4 and draw.data_ptr() % 16 == 0
        if self.kind == "canon":
            if (self.fused and NARROW_FUSED and aligned4 and grad.stride(0) == act.stride(0) and enc.stride(0) == 96
                    and not (grad.data_ptr() | act.data_ptr() | enc.data_ptr()) % 16):
                # the five narrow products below as ONE pass over the rows (csrc/backward_kernels.hip narrow5_kernel)
                _lib.check(L.swnerf_canon_narrow_grads(_lib.ptr(grad), grad.stride(0), _lib.ptr(act), act.stride(0), _lib.ptr(enc), _lib.ptr(draw), M,
                                                       _lib.ptr(s["c0s"]), _lib.ptr(s["cvs"]), _lib.ptr(s["gfeat"]), _lib.ptr(s["a4w"]), _lib.ptr(s["rgb4w"]),
                                                       _lib.ptr(g[1]), _lib.ptr(g[17]), _lib.ptr(s["a4b"]), _lib.ptr(s["rgb4b"]), _st(st)), "canon_narrow_grads")
                return
            cv = (s["cvs"], 0, 64, 32) if self.fused else (g[16], 256, Cpos, self.Cdir)
            mm(grad, 0, 256, enc, 0, e0, c0[0], c0[1], g[1])                               # pts_linears.0
            mm(grad, SW_ACT_HV, 128, act, SW_ACT_H7, 256, s["gfeat"], 0, g[17])           # G (+ views_linears.0.bias)
            mm(grad, SW_ACT_HV, 128, enc, cv[2], cv[3], cv[0], cv[1], None)               # views_linears.0, gamma(d) columns
            if aligned4:
                # all FOUR columns of d raw as output rows keep the operand 16-byte aligned with a column count that is a
                # multiple of 4 (the LDS-DMA staged GEMM variants): alpha_linear = row 3 of d raw^T . h7, rgb_linear = rows 0..2
                mm(draw, 0, 4, act, SW_ACT_H7, 256, s["a4w"], 0, s["a4b"])
                mm(draw, 0, 4, act, SW_ACT_HV, 128, s["rgb4w"], 0, s["rgb4b"])
            else:
                mm(draw, 3, 1, act, SW_ACT_H7, 256, s["a4w"], 3 * 256, s["a4b"][3:])      # row 3 of the 4-row form
                mm(draw, 0, 3, act, SW_ACT_HV, 128, s["rgb4w"], 0, s["rgb4b"])
        elif self.kind == "noview":
            # (the fused kernel exists for this net too but measures slower than its two skinny GEMMs - 304 vs 254 us per 393 216-row
            # chunk, profiles/r04/narrow_plan.md - so it is opt-in)
            if (self.fused and NOVIEW_NARROW_FUSED and draw.stride(0) == 8 and grad.stride(0) == act.stride(0) and enc.stride(0) == 96
                    and not (grad.data_ptr() | act.data_ptr() | enc.data_ptr() | draw.data_ptr()) % 16):
                _lib.check(L.swnerf_noview_narrow_grads(_lib.ptr(grad), grad.stride(0), _lib.ptr(act), act.stride(0), _lib.ptr(enc), _lib.ptr(draw), M,
                                                        _lib.ptr(s["c0s"]), _lib.ptr(s["w8"]), _lib.ptr(g[1]), _lib.ptr(s["b8"]), _st(st)), "noview_narrow_grads")
                return
            mm(grad, 0, 256, enc, 0, e0, c0[0], c0[1], g[1])                               # pts_linears.0
            mm(draw, 0, 8, act, SW_ACT_H7, 256, s["w8"], 0, s["b8"])                       # output_linear (rows 0..out_ch-1)
        else:
            if (self.fused and NARROW_FUSED and aligned4 and grad.stride(0) == act.stride(0) and enc.stride(0) == 96
                    and not (grad.data_ptr() | act.data_ptr() | enc.data_ptr()) % 16):
                _lib.check(L.swnerf_deform_narrow_grads(_lib.ptr(grad), grad.stride(0), _lib.ptr(act), act.stride(0), _lib.ptr(enc), _lib.ptr(draw), M,
                                                        _lib.ptr(s["c0s"]), _lib.ptr(s["cts"]), _lib.ptr(s["w4"]), _lib.ptr(g[1]), _lib.ptr(s["b4"]), _st(st)),
                           "deform_narrow_grads")
                return
            ct = (s["cts"], 0, enc, 64, 32) if self.fused else (g[0], Cpos, enc2, 0, self.Ct)
            mm(grad, 0, 256, enc, 0, e0, c0[0], c0[1], g[1])                               # _time.0 = [gamma(x) | gamma(t)]
            mm(grad, 0, 256, ct[2], ct[3], ct[4], ct[0], ct[1], None)
            ncol = 4 if aligned4 else draw.shape[1]
            mm(draw, 0, ncol, act, SW_ACT_H7, 256, s["w4"], 0, s["b4"] if ncol == 4 else s["b4"][:ncol])     # _time_out (rows 0..2)

    # -- after the last chunk ----------------------------------------------------------------------------------------------
    def finish(self, st):
        L, g, s = self.L, self.g, self.s
        Lp, Ld, Lt = self.bands
        if self.fused:                                           # slot-ordered columns to their reference columns
            jobs = [(s["c0s"], 64, 0, g[0], 0), (s["c5s"], 64, 0, g[10], 0)]
            if self.kind == "canon":
                jobs.append((s["cvs"], 32, 64, g[16], 256))
            for cs, nslots, slot0, W, col0 in jobs:
                _lib.check(L.swnerf_unslot_grad(_lib.ptr(cs), cs.stride(0), cs.shape[0], slot0, nslots, Lp, Ld if self.kind == "canon" else 0,
                                                W.data_ptr(), W.stride(0), col0, st), "unslot_grad")
            if self.kind == "deform":
                cts = s["cts"]
                _lib.check(L.swnerf_unslot_grad_time(_lib.ptr(cts), 32, 256, 32, Lt, g[0].data_ptr(), g[0].stride(0), self.Cpos, st), "unslot_grad_time")
        if self.kind == "canon":
            f32 = lambda p_: p_.detach() if (p_.dtype == torch.float32 and p_.is_contiguous()) else p_.detach().float().contiguous()
            Wv, W_f, b_f = f32(self.params[16]), f32(self.params[18]), f32(self.params[19])
            _lib.check(L.swnerf_feature_finish(_lib.ptr(s["gfeat"]), _lib.ptr(g[17]), _lib.ptr(Wv), Wv.stride(0), _lib.ptr(W_f), _lib.ptr(b_f),
                                               _lib.ptr(s["a4w"]), _lib.ptr(s["a4b"]), _lib.ptr(g[16]), g[16].stride(0), _lib.ptr(g[18]), _lib.ptr(g[19]),
                                               _lib.ptr(g[20]), _lib.ptr(g[21]), st), "feature_finish")
        return self.g
