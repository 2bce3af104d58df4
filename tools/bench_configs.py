#!/usr/bin/env python3
"""Throughput of the fused render path on the other BASELINE.json configs (single GPU).
Not the headline bench (bench.py = C2); writes a markdown table for profiles/.
C1 1024 rays x 64 coarse only | C2 4096 x (64+128) | C3 fern NDC 4096 x (64+128) |
C4 one GPU's shard of the 800x800 frame (80 000 rays, from get_rays_range) | C5 D-NeRF 20 000-ray shard, t=0.5 and t=0"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import synth, model, embedder, render, render_dnerf, ray

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
e10, c10 = embedder.get_embedder(10, 3, 0)
e4, c4 = embedder.get_embedder(4, 3, 0)
et, ct = embedder.get_embedder(10, 1, 0)


def static_net(spec):
    m = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(spec[0], alpha_bias=spec[1]).items()})
    return m.to(dev).eval()


coarse, fine = static_net(synth.NET_COARSE), static_net(synth.NET_FINE)
dn = model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=c10, output_ch=5, skips=[4], input_ch_views=c4,
                            input_ch_time=ct, use_viewdirs=True, embed_fn=e10, zero_canonical=True)
dn.load_state_dict({k: torch.from_numpy(v) for k, v in synth.dnerf_state_dict(synth.NET_DNERF[0], alpha_bias=synth.NET_DNERF[1]).items()})
dn = dn.to(dev).eval()
embed_fn, embeddirs_fn, embedtime_fn = e10, e4, et
q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
qd = lambda inputs, viewdirs, ts, network_fn: render_dnerf.run_network(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn,
                                                                       embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn)


def timeit(fn, flop_per_ray, n_rays, reps=10):
    with torch.no_grad():
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt * 1e3, n_rays / dt, n_rays * flop_per_ray / dt / 1e12


rows = []
K4, c2w4 = synth.lego_camera(400, 400)
o, d = synth.pick_rays(400, 400, K4, c2w4, 1024, 1)
kw = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=coarse, network_query_fn=q, white_bkgd=True, perturb=0., raw_noise_std=0.)
rows.append(("C1 1024 rays, 64 coarse only", ) + timeit(lambda: render.render(400, 400, K4, rays=(T(o), T(d)), N_samples=64, N_importance=0, **kw), 75956224, 1024, 50))
K8, c2w8 = synth.lego_camera(800, 800)
o, d = synth.pick_rays(800, 800, K8, c2w8, 4096, 2)
ro, rd = T(o), T(d)
rows.append(("C2 4096 rays, 64+128, two nets", ) + timeit(lambda: render.render(800, 800, K8, rays=(ro, rd), N_samples=64, N_importance=128, network_fine=fine, **kw), 303824896, 4096, 20))
kwp = dict(kw, perturb=1., raw_noise_std=1.0)
rows.append(("C2 with perturb=1, raw_noise_std=1 (torch RNG)", ) + timeit(lambda: render.render(800, 800, K8, rays=(ro, rd), N_samples=64, N_importance=128, network_fine=fine, **kwp), 303824896, 4096, 20))
Kf, c2wf = synth.fern_camera()
o, d = synth.pick_rays(378, 504, Kf, c2wf, 4096, 3)
fo, fd = T(o), T(d)
kwn = dict(kw, ndc=True, near=0., far=1., white_bkgd=False)
rows.append(("C3 fern NDC 4096 rays, 64+128", ) + timeit(lambda: render.render(378, 504, Kf, rays=(fo, fd), N_samples=64, N_importance=128, network_fine=fine, **kwn), 303824896, 4096, 20))
lo, hi = synth.shard_range(800 * 800, 8, 3)


def c4():
    so, sd = ray.get_rays_range(800, 800, K8, c2w8, lo, hi - lo, dev)
    return render.render(800, 800, K8, chunk=1 << 30, rays=(so, sd), N_samples=64, N_importance=128, network_fine=fine, **kw)


rows.append(("C4 one GPU's shard of the 800x800 frame: 80 000 rays incl. get_rays", ) + timeit(c4, 303824896, hi - lo, 3))
lo5, hi5 = synth.shard_range(400 * 400, 8, 3)
kwd = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=dn, network_query_fn=qd, white_bkgd=True, perturb=0.,
           raw_noise_std=0., N_samples=64, N_importance=128)
for tv, flop in ((0.5, 558366720), (0.0, 303824896)):
    def c5(tv=tv):
        so, sd = ray.get_rays_range(400, 400, float(K4[0, 0]), c2w4, lo5, hi5 - lo5, dev)
        return render_dnerf.render(400, 400, float(K4[0, 0]), chunk=1 << 30, rays=(so, sd), frame_time=tv, **kwd)
    rows.append((f"C5 D-NeRF shard 20 000 rays, 64+128, one net, t={tv}", ) + timeit(c5, flop, hi5 - lo5, 5))
print("| config (1x MI355X, fp32) | ms / call | rays/s | algorithmic TFLOP/s | % of 157.3 |")
print("|---|---|---|---|---|")
for name, ms, rps, tf in rows:
    print(f"| {name} | {ms:.2f} | {rps:,.0f} | {tf:.1f} | {100*tf/157.3:.1f} |")
