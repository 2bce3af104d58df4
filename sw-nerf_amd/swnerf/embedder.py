"""Drop-in for the reference's embedder.py (embedder.py:1-59): same names, same call
signatures, sinusoidal positional encoding evaluated by the HIP kernel `swnerf_embed`.

`get_embedder` returns a callable OBJECT (not a bare lambda) carrying `.multires` and
`.input_dims`, so render_rays can recognise the standard embedders inside the
`network_query_fn` closure that create_nerf builds (nerf/run.py:248-251) and switch to the
fused kernel, where the encoding never leaves registers."""
import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (re-exported: run_dnerf.py gets torch/nn/F/np via star-import)
import numpy as np

from . import _lib

img2mse = lambda x, y: torch.mean((x - y) ** 2)
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.Tensor([10.]).to(x.device))
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


def _embed_hip(x, multires):
    x = _lib.dev_f32(x, "inputs")
    d = x.shape[-1]
    flat = x.reshape(-1, d)
    out = torch.empty((flat.shape[0], d * (1 + 2 * multires)), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().swnerf_embed(_lib.ptr(flat), flat.shape[0], d, multires, _lib.ptr(out),
                                       _lib.stream_of(x)), "embed")
    return out.reshape(*x.shape[:-1], out.shape[-1])


class Embedder:
    """embedder.py:12-42.  Only the configuration the reference ever builds is accelerated:
    include_input, log_sampling, periodic_fns=[sin, cos], max_freq_log2 = num_freqs-1."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()

    def create_embedding_fn(self):
        kw = self.kwargs
        d = kw['input_dims']
        L = kw['num_freqs']
        fns = list(kw.get('periodic_fns', []))
        ok = (kw.get('include_input', False) and kw.get('log_sampling', False)
              and len(fns) == 2 and fns[0] is torch.sin and fns[1] is torch.cos
              and kw['max_freq_log2'] == L - 1 and 0 <= L <= 24)
        if not ok:
            raise NotImplementedError(
                "swnerf.Embedder: only include_input=True, log_sampling=True, periodic_fns=[torch.sin, torch.cos], "
                "max_freq_log2=num_freqs-1 is built as a HIP kernel (the only configuration get_embedder creates)")
        self.multires = L
        self.input_dims = d
        self.out_dim = d * (1 + 2 * L)

    def embed(self, inputs):
        return _embed_hip(inputs, self.multires)


class EmbedFn:
    """What get_embedder returns in place of the reference's lambda (embedder.py:58)."""

    def __init__(self, eo):
        self.eo = eo
        self.multires = eo.multires
        self.input_dims = eo.input_dims
        self.out_dim = eo.out_dim

    def __call__(self, x):
        return self.eo.embed(x)


class _Identity(nn.Identity):
    multires = -1


def get_embedder(multires, input_dims, i=0):
    """embedder.py:44-59."""
    if i == -1:
        return _Identity(), input_dims
    eo = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1,
                  num_freqs=multires, log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return EmbedFn(eo), eo.out_dim
