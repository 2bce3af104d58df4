#!/usr/bin/env python3
"""Capture golden vectors G12 from the UNMODIFIED reference (build container only): render_rays of nerf/run.py with
`use_viewdirs=False` - the reference's argparse default (utils.py:43; model.py:59-60; 8-column ray batch nerf/run.py:152-157)
- on NON-degenerate weights (the G11 end-to-end case renders fully transparent, which pins nothing of the image): two
8x256 nets without the view branch, output_ch = 5 (nerf/run.py:231), 256 lego-like rays, 64 coarse samples alone and
64 + 128 hierarchical.  Only OUTPUTS (plus a checksum of the seeded inputs) are stored.
Run: python tests/golden/make_golden_noview.py"""
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
g = cases.g12_inputs()
nets = []
for sd in cases.g12_weights():
    net = MODEL.vallina_NeRF(**cases.G12_NET)
    net.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    nets.append(net)
embed_fn, _ = EMB.get_embedder(10, 3, 0)
embeddirs_fn = None                                                                                  # nerf/run.py:227-229
query = lambda inputs, viewdirs, network_fn: RUN.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                             embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
o_, d_ = T(g["rays_o"]), T(g["rays_d"])
rb = torch.cat([o_, d_, g["near"] * torch.ones_like(d_[:, :1]), g["far"] * torch.ones_like(d_[:, :1])], -1)     # nerf/run.py:152-154
out = {"checksum": cases.checksum(g["rays_o"], g["rays_d"], *[v for sd in cases.g12_weights() for v in sd.values()])}
ret = RUN.render_rays(rb, nets[0], query, 64, retraw=True, N_importance=0, white_bkgd=True, perturb=0., raw_noise_std=0.)
for k in ("rgb_map", "disp_map", "acc_map"):
    out[f"c_{k}"] = ret[k].numpy()
out["c_raw"] = ret["raw"][:16].numpy()
ret = RUN.render_rays(rb, nets[0], query, 64, retraw=True, N_importance=128, network_fine=nets[1], white_bkgd=True, perturb=0., raw_noise_std=0.)
for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"):
    out[f"h_{k}"] = ret[k].numpy()
out["h_raw_shape"] = np.asarray(ret["raw"].shape)
np.savez_compressed(os.path.join(HERE, "g12_noview.npz"), **out)
print("wrote g12_noview.npz:", {k: v.shape for k, v in out.items()})
print("acc coarse: mean %.3f min %.3f max %.3f; acc fine: mean %.3f" % (out["c_acc_map"].mean(), out["c_acc_map"].min(), out["c_acc_map"].max(), out["h_acc_map"].mean()))
