"""The NeRF MLPs at ANY shape (model.py:10-62, 93-151, 227-296): layer by layer on the tiled fp32-MFMA GEMM of
csrc/generic_kernels.hip.  Taken when a module is not the shape the fused kernels are built for (D=8, W=256, skips=[4],
use_viewdirs=True with get_embedder-sized inputs): `use_viewdirs=False` - the reference's argparse default,
utils.py:26-29, handled at model.py:59-60 -, other depths / widths / skip sets, other input sizes.  Slower than the
fused path (every activation round-trips HBM, like in the reference), same arithmetic class (fp32 MFMA, fp32
accumulate), differentiable: dX = dY.W (swnerf_gemm_nn), dW = dY^T.X / db (swnerf_gemm_tn), relu' (swnerf_relu_mask).
The `cat`/`split` glue between layers is torch (memory movement only)."""
import torch

from . import _lib


def _c32(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _linear_raw(x, weight, bias, relu):
    L = _lib.lib()
    M, K = x.shape
    N = weight.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _lib.check(L.swnerf_linear(_lib.ptr(x), x.stride(0), M, K, _lib.ptr(weight), _lib.ptr(bias), N, int(relu), _lib.ptr(y), N,
                               _lib.stream_of(x)), "linear")
    return y


class _Linear(torch.autograd.Function):
    """y = act(x . W^T + b) with hand-written forward and backward kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        y = _linear_raw(x, weight, bias, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else x.new_empty(0))
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        L = _lib.lib()
        st = _lib.stream_of(x)
        M, K = x.shape
        N = weight.shape[0]
        dy = _c32(dy)
        if ctx.relu:
            dy = dy.clone()                                  # masked in place; the caller's tensor stays untouched
            _lib.check(L.swnerf_relu_mask(_lib.ptr(dy), _lib.ptr(y), dy.numel(), st), "relu_mask")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=x.device)
            _lib.check(L.swnerf_gemm_nn(_lib.ptr(dy), N, M, N, _lib.ptr(weight), K, K, _lib.ptr(dx), K, st), "gemm_nn")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.zeros((N, K), dtype=torch.float32, device=x.device)
            db = torch.zeros((N,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
            for o0 in range(0, N, 256):                      # swnerf_gemm_tn: up to 256 output rows per call
                no = min(256, N - o0)
                _lib.check(L.swnerf_gemm_tn(dy.data_ptr() + 4 * o0, N, no, _lib.ptr(x), x.stride(0), K, M, dw.data_ptr() + 4 * o0 * K, K,
                                            (db.data_ptr() + 4 * o0) if db is not None else None, st), "gemm_tn")
        return dx, dw, db, None


def linear(x, lin, relu=False):
    """nn.Linear `lin` applied to x [M, K] by the HIP GEMM (+ relu fused)."""
    x = _lib.dev_f32(x, "x", lin.in_features)
    w, b = lin.weight, lin.bias
    if not w.is_cuda:
        raise RuntimeError("swnerf: module parameters must be on the GPU (call .to('cuda')); no CPU fallback")
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        return _Linear.apply(x, _c32(w), None if b is None else _c32(b), bool(relu))
    return _linear_raw(x, _c32(w.detach()), None if b is None else _c32(b.detach()), bool(relu))


def canonical_forward(mod, x):
    """vallina_NeRF.forward / NeRFOriginal.forward (model.py:39-62, 273-296) for any D, W, skips, use_viewdirs."""
    lead = x.shape[:-1]
    x = _lib.dev_f32(x, "x").reshape(-1, x.shape[-1])
    input_pts, input_views = torch.split(x, [mod.input_ch, mod.input_ch_views], dim=-1)
    h = input_pts
    for i, l in enumerate(mod.pts_linears):
        h = linear(h, l, relu=True)
        if i in mod.skips:
            h = torch.cat([input_pts, h], -1)
    if mod.use_viewdirs:
        alpha = linear(h, mod.alpha_linear)
        feature = linear(h, mod.feature_linear)
        h = torch.cat([feature, input_views], -1)
        for l in mod.views_linears:
            h = linear(h, l, relu=True)
        outputs = torch.cat([linear(h, mod.rgb_linear), alpha], -1)
    else:
        outputs = linear(h, mod.output_linear)
    return outputs.reshape(*lead, outputs.shape[-1])


class _Embed(torch.autograd.Function):
    """Embedder.embed (embedder.py:33-42) with its input gradient: d sin(2^k x) = 2^k cos(2^k x), d cos = -2^k sin -
    both already sit in the saved output, so the backward is a handful of elementwise ops on it."""

    @staticmethod
    def forward(ctx, x, multires):
        from .embedder import _embed_hip
        out = _embed_hip(x, multires)
        ctx.L, ctx.d = multires, x.shape[-1]
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        d, Lb = ctx.d, ctx.L
        gx = g[..., :d].clone()
        for k in range(Lb):
            s0 = d * (1 + 2 * k)
            sin_k, cos_k = out[..., s0:s0 + d], out[..., s0 + d:s0 + 2 * d]
            gx = gx + float(2 ** k) * (g[..., s0:s0 + d] * cos_k - g[..., s0 + d:s0 + 2 * d] * sin_k)
        return gx, None


def embed_with_grad(embed_fn, x):
    """embed_fn(x); through `_Embed` when x carries a gradient and embed_fn is this build's encoder."""
    from .embedder import EmbedFn
    if isinstance(embed_fn, EmbedFn) and torch.is_grad_enabled() and x.requires_grad:
        return _Embed.apply(_c32(x), embed_fn.multires)
    return embed_fn(x)


def temporal_forward(mod, x, ts):
    """DirectTemporalNeRF.forward (model.py:128-151) for any shape: deformation net -> x + dx -> re-embed -> `_occ`."""
    lead = x.shape[:-1]
    x = _lib.dev_f32(x, "x").reshape(-1, x.shape[-1])
    input_pts, input_views = torch.split(x, [mod.input_ch, mod.input_ch_views], dim=-1)
    t = ts[0].reshape(-1, ts[0].shape[-1])
    lo, hi = torch.aminmax(t[:, :1])
    lo, hi = float(lo), float(hi)
    assert lo == hi, "Only accepts all points from same time"                # model.py:141
    if lo == 0. and mod.zero_canonical:
        dx = torch.zeros_like(input_pts[:, :3])
    else:
        h = torch.cat([input_pts, _lib.dev_f32(t, "ts[0]")], dim=-1)         # query_time, model.py:128-136
        for i, l in enumerate(mod._time):
            h = linear(h, l, relu=True)
            if i in mod.skips:
                h = torch.cat([input_pts, h], -1)
        dx = linear(h, mod._time_out)
        if mod.embed_fn is None:
            raise RuntimeError("swnerf: DirectTemporalNeRF needs embed_fn to re-embed x + dx (model.py:148-149)")
        input_pts = embed_with_grad(mod.embed_fn, input_pts[:, :3] + dx)
    out = canonical_forward(mod._occ, torch.cat([input_pts, input_views], dim=-1))
    return out.reshape(*lead, out.shape[-1]), dx.reshape(*lead, 3)
