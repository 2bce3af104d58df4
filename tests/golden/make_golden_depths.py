#!/usr/bin/env python3
"""Capture golden G7z from the UNMODIFIED reference (build container only): the fine-pass DEPTHS of the G7 C2 case
(1024 lego-like rays, 64 + 128, two nets).  The reference's static render_rays does not return them (nerf/run.py:405-416),
so the call is the same as for g7_c2.npz and the depths are taken where the reference itself produces them:
`z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)` (nerf/run.py:400) - torch.sort is wrapped for the duration
of the call and its result recorded; the reference runs unmodified.  The script first checks that this run reproduces the
committed g7_c2.npz bit for bit, so the depths belong to exactly that render.  Stored: z_vals of the first 32 rays (the rays
whose `raw` g7_c2 holds).   Run: python tests/golden/make_golden_depths.py"""
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
e10, _ = EMB.get_embedder(10, 3, 0)
e4, _ = EMB.get_embedder(4, 3, 0)
nets = []
for sd in cases.weights_static():
    m = MODEL.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: T(v) for k, v in sd.items()})
    nets.append(m)
q = lambda inputs, viewdirs, network_fn: RUN.run_network(inputs, viewdirs, network_fn, embed_fn=e10, embeddirs_fn=e4, netchunk=1024 * 64)
g = cases.g7_inputs()
o, d = T(g["rays_o"]), T(g["rays_d"])
rb = torch.cat([o, d, g["near"] * torch.ones_like(d[:, :1]), g["far"] * torch.ones_like(d[:, :1]), d / torch.norm(d, dim=-1, keepdim=True)], -1)
seen = []
real_sort = torch.sort


def recording_sort(*a, **k):
    out = real_sort(*a, **k)
    seen.append(out[0].clone())
    return out


torch.sort = recording_sort
try:
    r = RUN.render_rays(rb, nets[0], q, 64, retraw=True, N_importance=128, network_fine=nets[1], white_bkgd=True)
finally:
    torch.sort = real_sort
assert len(seen) == 1 and tuple(seen[0].shape) == (1024, 192), [tuple(s.shape) for s in seen]
committed = dict(np.load(os.path.join(HERE, "g7_c2.npz")))
for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "z_std"):
    assert np.array_equal(r[k].numpy(), committed[k], equal_nan=True), f"this run does not reproduce the committed g7_c2.npz ({k})"
assert np.array_equal(r["raw"][:32].numpy(), committed["raw"])
np.savez_compressed(os.path.join(HERE, "g7_c2_depths.npz"), z_vals=seen[0][:32].numpy(), crc=cases.checksum(g["rays_o"], g["rays_d"]))
print("wrote g7_c2_depths.npz: z_vals", tuple(seen[0][:32].shape), "- the run reproduced g7_c2.npz bit for bit")
