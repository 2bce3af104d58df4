#!/usr/bin/env python3
"""Soak of the D-NeRF inference dispatch (round 4): random ray counts, sample counts, frame times (incl. t = 0, the canonical-only
branch), lindisp / perturb / noise / white background - the FUSED path (deformation + canonical net in one launch per pass, layer 0
of the deformation net started from the per-ray TIME tile) against the op-by-op path of the same library (mlp_forward per row,
TIME in line).  Both return their fine depths (`z_vals`, run_dnerf.py:476), so every difference is attributed:
  * a ray whose depths are bit-equal on both paths must agree to 2e-5 in rgb, and where the sample is the same its raw / dx must be
    the same BITS (the two paths run the same additions in the same order);
  * rays with a moved depth are counted (none are expected: the coarse passes are bit-identical) and bounded by one coarse interval.
usage: soak_dnerf.py [seconds]   (run under `timeout`)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import synth, model, render, render_dnerf, embedder

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "11")))
embed_fn, _ = embedder.get_embedder(10, 3, 0)
embeddirs_fn, _ = embedder.get_embedder(4, 3, 0)
embedtime_fn, _ = embedder.get_embedder(10, 1, 0)
qd = lambda inputs, viewdirs, ts, network_fn: render_dnerf.run_network(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                                                        embedtime_fn=embedtime_fn, netchunk=1024 * 64, embd_time_discr=True)
opaque = lambda a, b, c, d, _q=qd: _q(a, b, c, d)
nets = []
for seed, ab in ((synth.NET_DNERF[0], synth.NET_DNERF[1]), (4242, -0.5)):
    m = model.DirectTemporalNeRF(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, output_ch=5, skips=[4], use_viewdirs=True,
                                 embed_fn=embed_fn, zero_canonical=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.dnerf_state_dict(seed, alpha_bias=ab).items()})
    nets.append(m.to(dev).eval())
K, c2w = synth.lego_camera(400, 400)
t_end, it, worst, worst_tag, n_moved, n_rays, n_same = time.time() + budget, 0, -1.0, "", 0, 0, 0
with torch.no_grad():
    while time.time() < t_end:
        it += 1
        N = int(rng.choice([1, 2, 3, 5, 37, 256, 1023, 1024, 1025, int(rng.integers(1, 2000))]))
        S = int(rng.choice([3, 7, 32, 33, 64, 100, 192, int(rng.integers(3, 200))]))
        Ni = int(rng.choice([0, 0, 1, 5, 64, 128, 129, int(rng.integers(1, 200))]))
        tv = float(rng.choice([0.0, 0.25, 0.5, 1.0, float(rng.random())]))
        kw = dict(white_bkgd=bool(rng.integers(2)), lindisp=bool(rng.integers(2)), perturb=float(rng.integers(2)), pytest=True,
                  raw_noise_std=float(rng.choice([0.0, 0.0, 1.0])), retraw=True)
        net = nets[int(rng.integers(2))]
        seed = int(rng.integers(1 << 30))
        o, d = synth.pick_rays(400, 400, K, c2w, N, seed)
        rb = render.pack_ray_batch(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 2., 6., frame_time=tv)
        tag = f"it {it}: N={N} S={S} Ni={Ni} t={tv:.3f} seed={seed} {kw}"
        a = render_dnerf.render_rays(rb, net, qd, S, N_importance=Ni, **kw)
        b = render_dnerf.render_rays(rb, net, opaque, S, N_importance=Ni, **kw)
        assert list(a.keys()) == list(b.keys()), (tag, list(a.keys()), list(b.keys()))
        for k in a:
            assert a[k].shape == b[k].shape, (tag, k)
        rgb, acc = a["rgb_map"], a["acc_map"]
        assert bool(torch.isfinite(rgb).all()) and float(acc.min()) >= -1e-5 and float(acc.max()) <= 1 + 1e-4, tag
        za, zb = a["z_vals"], b["z_vals"]
        same = (za == zb).all(-1)
        drgb = (rgb - b["rgb_map"]).abs().amax(-1)
        if bool(same.any()):
            assert float(drgb[same].max()) <= 2e-5, (tag, "rays with bit-equal depths differ in rgb", float(drgb[same].max()))
            assert torch.equal(a["position_delta"][same], b["position_delta"][same]), (tag, "dx differs at equal depths")
            assert torch.equal(a["raw"][same], b["raw"][same]) or kw["raw_noise_std"] > 0 and float((a["raw"][same] - b["raw"][same]).abs().max()) <= 1e-6, (tag, "raw differs at equal depths")
        moved = int((~same).sum())
        if moved:
            t_rand = None
            if kw["perturb"] > 0:
                np.random.seed(0)
                t_rand = torch.Tensor(np.random.rand(N, S)).to(dev)
            zc = render.sample_coarse(rb, S, kw["lindisp"], t_rand)
            width = (zc[:, 1:] - zc[:, :-1]).amax(-1)
            assert bool(((za - zb).abs().amax(-1) <= width * 1.0000001)[~same].all()), (tag, "a depth moved by more than one coarse interval")
            assert moved <= max(2, int(0.1 * N)), (tag, "too many rays with a moved depth", moved, N)
        n_moved, n_rays, n_same = n_moved + moved, n_rays + N, n_same + int(same.sum())
        dmax = float(drgb.max())
        if dmax > worst:
            worst, worst_tag = dmax, tag
        if N >= 5:                                                       # rays are independent: a sub-batch renders to the same bits
            kw0 = dict(kw, perturb=0., raw_noise_std=0.)
            sub = render_dnerf.render_rays(rb[2:5].contiguous(), net, qd, S, N_importance=Ni, **kw0)
            full = render_dnerf.render_rays(rb, net, qd, S, N_importance=Ni, **kw0)
            assert torch.equal(torch.nan_to_num(sub["rgb_map"], nan=-7.), torch.nan_to_num(full["rgb_map"][2:5], nan=-7.)), tag
        if it % 25 == 0:
            print(f"{it} cases ok, worst |d rgb| fused vs op path so far {worst:.2e}; rays with a moved depth {n_moved} of {n_rays}", flush=True)
print(f"worst case: {worst_tag}")
print(f"soak_dnerf: {it} random cases, all ok; worst |d rgb| fused vs op path {worst:.2e}; rays with a moved depth: {n_moved} of {n_rays} "
      f"(raw and dx bit-identical on the {n_same} rays with equal depths)")
