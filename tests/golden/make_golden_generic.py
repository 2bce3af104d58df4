#!/usr/bin/env python3
"""Capture golden vectors G11 from the UNMODIFIED reference (build container only): the MLPs at shapes other than
the shipped 8x256/use_viewdirs one - `use_viewdirs=False` (the argparse default, utils.py:26-29; model.py:59-60),
other depths / widths / skip sets, DirectTemporalNeRF at D=4 - and one end-to-end render_rays without view directions.
Only OUTPUTS (plus a checksum of the seeded inputs) are stored.   Run: python tests/golden/make_golden_generic.py"""
import os
import sys
import types
import importlib
import importlib.util
import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

REF = "/root/reference"
for name in ["imageio", "lpips", "skimage", "skimage.metrics", "cv2", "configargparse", "torch.utils.tensorboard"]:
    try:
        importlib.import_module(name)
    except Exception:
        sys.modules[name] = types.ModuleType(name)
for attr in ("peak_signal_noise_ratio", "structural_similarity"):
    if not hasattr(sys.modules["skimage.metrics"], attr):
        setattr(sys.modules["skimage.metrics"], attr, None)
if not hasattr(sys.modules["torch.utils.tensorboard"], "SummaryWriter"):
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object

import torch  # noqa: E402
sys.path.insert(0, REF)
import embedder as EMB     # noqa: E402
import model as MODEL      # noqa: E402


def _load(path, name):
    cwd = os.getcwd()
    os.chdir(os.path.dirname(path))
    sys.path.insert(0, os.path.dirname(path))
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        os.chdir(cwd)
        sys.path.pop(0)
    return mod


RUN = _load(os.path.join(REF, "nerf", "run.py"), "ref_nerf_run")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
torch.set_grad_enabled(False)
g = cases.g11_inputs()
e10, _ = EMB.get_embedder(10, 3, 0)
e6, _ = EMB.get_embedder(6, 3, 0)
e4, _ = EMB.get_embedder(4, 3, 0)
et, _ = EMB.get_embedder(10, 1, 0)
out = {"checksum": cases.checksum(g["pts"], g["dirs"], g["rays"]["rays_o"], g["rays"]["rays_d"])}
for name, kw in cases.G11_NETS.items():
    net = MODEL.vallina_NeRF(**kw)
    net.load_state_dict({k: T(v) for k, v in cases.g11_weights(name).items()}, strict=True)
    emb = e10 if kw["input_ch"] == 63 else e6
    x = emb(T(g["pts"]))
    if kw["input_ch_views"]:
        x = torch.cat([x, e4(T(g["dirs"]))], -1)
    out[f"mlp_{name}"] = net(x).numpy()
kw = dict(cases.G11_DNERF)
dn = MODEL.NeRF.get_by_name("direct_temporal", embed_fn=e10, zero_canonical=True, **kw)
dn.load_state_dict({k: T(v) for k, v in cases.g11_dnerf_weights().items()}, strict=True)
x = torch.cat([e10(T(g["pts"])), e4(T(g["dirs"]))], -1)
for tv in (0.0, 0.5):
    te = et(torch.full((x.shape[0], 1), tv))
    o, dx = dn(x, [te, te])
    out[f"dnerf_out_t{int(tv * 10)}"], out[f"dnerf_dx_t{int(tv * 10)}"] = o.numpy(), dx.numpy()
# end to end: render_rays of nerf/run.py with use_viewdirs=False (ray batch of 8 columns, output_ch 5, one net for both passes)
net = MODEL.vallina_NeRF(**cases.G11_NETS["novd"])
net.load_state_dict({k: T(v) for k, v in cases.g11_weights("novd").items()})
embed_fn, embeddirs_fn = e10, None
query = lambda inputs, viewdirs, network_fn: RUN.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                             embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
r = g["rays"]
o_, d_ = T(r["rays_o"]), T(r["rays_d"])
rb = torch.cat([o_, d_, 2. * torch.ones_like(d_[:, :1]), 6. * torch.ones_like(d_[:, :1])], -1)     # nerf/run.py:152-154
ret = RUN.render_rays(rb, net, query, 32, N_importance=32, network_fine=None, white_bkgd=True, perturb=0., raw_noise_std=0.)
for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"):
    out[f"rr_{k}"] = ret[k].numpy()
np.savez_compressed(os.path.join(HERE, "g11_generic.npz"), **out)
print("wrote g11_generic.npz:", {k: v.shape for k, v in out.items()})
