#!/usr/bin/env python3
"""Round-3 soak of the inference dispatch: random ray counts, sample counts on both sides of the fused pass's LDS limits
(256 coarse / 1024 merged), nets with and without view directions, lindisp / perturb / noise / white background on and
off - the FUSED path (one or two `swnerf_render_pass` launches, or the fused pass + resampling ops beyond the limits)
against the op-by-op path of the same library (an opaque closure hides the encoders), plus invariants: finite colours,
acc in [0, 1], NaN disparity only on empty rays, a sub-batch renders to the same bits.
usage: soak_r03.py [seconds]   (run under `timeout`)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import synth, model, render, embedder

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "3")))
embed_fn, c10 = embedder.get_embedder(10, 3, 0)
e4, c4 = embedder.get_embedder(4, 3, 0)


def nets(views):
    out = []
    for k in range(2):
        if views:
            sd = synth.nerf_state_dict(7000 + k, alpha_bias=(-0.25, -1.0)[k])
            m = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[4], use_viewdirs=True)
        else:
            sd = synth.noview_state_dict(7100 + k, alpha_bias=(0.5, 0.7)[k], output_ch=(5, 4)[k])
            m = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=0, output_ch=(5, 4)[k], skips=[4], use_viewdirs=False)
        m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()})
        out.append(m.to(dev).eval())
    return out


NETS = {True: nets(True), False: nets(False)}
K, c2w = synth.lego_camera(400, 400)
t_end, it, worst = time.time() + budget, 0, 0.0
with torch.no_grad():
    while time.time() < t_end:
        it += 1
        views = bool(rng.integers(2))
        embeddirs_fn = e4 if views else None
        q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
        opaque = lambda a, b, c, _q=q: _q(a, b, c)
        N = int(rng.choice([1, 2, 3, 5, 37, 256, 1023, 1024, 1025, int(rng.integers(1, 3000))]))
        S = int(rng.choice([2, 3, 7, 32, 33, 64, 100, 256, 257, 300, int(rng.integers(2, 320))]))
        Ni = int(rng.choice([0, 0, 1, 5, 64, 128, 129, 700, 900, int(rng.integers(1, 200))]))
        if S < 3:
            Ni = 0                                             # the reference itself cannot resample 2 coarse samples
        kw = dict(white_bkgd=bool(rng.integers(2)), lindisp=bool(rng.integers(2)), perturb=float(rng.integers(2)), pytest=True,
                  raw_noise_std=float(rng.choice([0.0, 0.0, 1.0])), retraw=bool(rng.integers(2)))
        two = bool(rng.integers(2))
        o, d = synth.pick_rays(400, 400, K, c2w, N, int(rng.integers(1 << 30)))
        rb = render.pack_ray_batch(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 2., 6.)
        if not views:
            rb = rb[:, :8].contiguous()
        n0, n1 = NETS[views]
        assert render.fused_plan(q, [n0, n1 if two else None]) is not None and render.fused_plan(opaque, [n0]) is None
        a = render.render_rays(rb, n0, q, S, N_importance=Ni, network_fine=n1 if two else None, **kw)
        b = render.render_rays(rb, n0, opaque, S, N_importance=Ni, network_fine=n1 if two else None, **kw)
        assert list(a.keys()) == list(b.keys()), (list(a.keys()), list(b.keys()))
        tag = f"it {it}: views={views} N={N} S={S} Ni={Ni} two={two} {kw}"
        for k in a:
            assert a[k].shape == b[k].shape, (tag, k, a[k].shape, b[k].shape)
        rgb, acc, disp = a["rgb_map"], a["acc_map"], a["disp_map"]
        assert bool(torch.isfinite(rgb).all()) and float(acc.min()) >= -1e-5 and float(acc.max()) <= 1 + 1e-4, tag
        assert bool((torch.isnan(disp) == (acc == 0)).all()), tag
        # same arithmetic both ways up to the resampling's own conditioning: nearly every pixel agrees to 2e-5
        dlt = (rgb - b["rgb_map"]).abs()
        frac = float((dlt <= 2e-5).float().mean())
        worst = max(worst, float(dlt.max()))
        # (hundreds of drawn samples per ray: the chance that ONE of them sits at a flipping bin edge of sample_pdf - DESIGN.md 6 -
        # grows with their number; 65 % of the rays within 2e-5 at 900 samples, max 1e-2, measured)
        # The gates of the golden cases (tests/test_gpu_parity.py _cmp): >= 70 % within 2e-5, >= 90 % within 2e-4.
        frac4 = float((dlt <= 2e-4).float().mean())
        lo5, lo4 = ((0.7, 0.9) if Ni <= 200 else (0.5, 0.7)) if N >= 37 else (0.0, 0.0)
        # a flipped sample moves by up to one bin: few coarse samples, lindisp spacing (wide far bins) or hundreds of draws widen the tail
        hard = 5e-2 if (S >= 64 and not kw["lindisp"] and Ni <= 200) else (0.25 if S >= 32 else 1.0)
        assert frac >= lo5 and frac4 >= lo4 and float(dlt.max()) <= hard, (tag, frac, frac4, float(dlt.max()))
        if "rgb0" in a:
            assert float((a["rgb0"] - b["rgb0"]).abs().max()) <= 2e-5, tag                      # in front of the resampling: tight
        if N >= 5:                                                                               # rays are independent
            sub = render.render_rays(rb[2:5].contiguous(), n0, q, S, N_importance=Ni, network_fine=n1 if two else None,
                                     **dict(kw, perturb=0., raw_noise_std=0.))
            full = render.render_rays(rb, n0, q, S, N_importance=Ni, network_fine=n1 if two else None, **dict(kw, perturb=0., raw_noise_std=0.))
            assert torch.equal(torch.nan_to_num(sub["rgb_map"], nan=-7.), torch.nan_to_num(full["rgb_map"][2:5], nan=-7.)), tag
        if it % 25 == 0:
            print(f"{it} cases ok, worst |d rgb| fused vs op path so far {worst:.2e}", flush=True)
print(f"soak_r03: {it} random cases, all ok; worst |d rgb| fused vs op path {worst:.2e}")
