"""Random shapes through the fused training passes (with and without view directions), every parameter gradient against
the float64 evaluation of the oracle with exact ReLU-flip accounting (tests/flipcheck.py): ray counts that are not multiples
of 4, sample counts that are not multiples of 32, lindisp, stratified jitter, raw noise, white background, 4 / 5 output
channels, backward chunks of a few rays, a gradient on every output.  Seeded: the same cases every run."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import nerf_oracle as O
from flipcheck import flip_aware_check

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("views", [False, True])
def test_random_training_shapes(views, monkeypatch):
    import swnerf.embedder as embedder, swnerf.render as render, swnerf.model as model
    dev = torch.device("cuda:0")
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn = embedder.get_embedder(4, 3, 0)[0] if views else None
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    rng = np.random.default_rng(4100 + int(views))
    sds = cases.weights_static() if views else cases.g12_weights()

    def mk(i, oc):
        sd = dict(sds[i])
        if views:
            m = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        else:
            sd["output_linear.weight"], sd["output_linear.bias"] = sd["output_linear.weight"][:oc], sd["output_linear.bias"][:oc]
            m = model.vallina_NeRF(**dict(cases.G12_NET, output_ch=oc))
        m.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
        return m.to(dev).train(), sd

    used = lambda net: {k: p.grad for k, p in net.named_parameters() if views or k.startswith(("pts_linears", "output_linear"))}
    worst_flips = 0
    with torch.enable_grad():
        for case in range(int(os.environ.get("SWNERF_TRAIN_RANDOM_CASES", "10"))):     # a soak: set it to a few hundred
            n, S = int(rng.integers(1, 41)), int(rng.choice([2, 3, 17, 31, 32, 33, 48, 64, 65, 96, 127]))
            hier = case % 10 >= 6
            Ni = int(rng.choice([1, 16, 40, 128])) if hier else 0
            S = max(S, 3) if hier else S
            if S + Ni > 256:
                Ni = 256 - S
            oc = 4 if views else int(rng.choice([4, 5]))
            white, lindisp, jitter = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
            noise_std = float(rng.choice([0., 0.7]))
            monkeypatch.setattr(render, "TRAIN_BWD_CHUNK_ROWS", int(rng.choice([64, 1024, 393216])))
            g = cases.g7_inputs(n=n, seed=5000 + 17 * case + int(views))
            rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6.)
            rb = (rb if views else rb[:, :8]).contiguous()
            tgt = T(rng.uniform(0, 1, (n, 3)).astype(np.float32))
            wa, wd = T(rng.standard_normal(n).astype(np.float32)), T(rng.standard_normal(n).astype(np.float32))
            S1 = S + Ni
            Graw = T((1e-3 * rng.standard_normal((n, S1, oc))).astype(np.float32))

            def ray_loss(ret, idx, key="rgb_map"):            # a sum over rays: img2mse + terms on disp, acc and raw
                c = lambda t: t[idx.to(t.device)].to(ret[key])
                L = ((ret[key] - c(tgt)) ** 2).sum() / (3 * n)
                if key == "rgb_map":
                    ok = ~torch.isnan(ret["disp_map"])
                    L = L + 0.1 * (ret["acc_map"] * c(wa)).sum() / n + (ret["raw"] * c(Graw)).sum() \
                        + 0.01 * (torch.where(ok, ret["disp_map"], torch.zeros_like(ret["disp_map"])) * c(wd)).sum() / n
                return L

            kw = dict(retraw=True, N_importance=Ni, white_bkgd=white, lindisp=lindisp, perturb=1. if jitter else 0., raw_noise_std=noise_std, pytest=True)
            (nc, sd_c), (nf, sd_f) = mk(0, oc), mk(1, oc)
            hits = []
            monkeypatch.setattr(render, "PASS_HOOK", lambda *a: hits.append(a))
            ret = render.render_rays(rb.to(dev), nc, q, S, network_fine=nf if hier else None, **kw)
            monkeypatch.setattr(render, "PASS_HOOK", None)
            assert len(hits) == (4 if hier else 2), hits                        # begin / end of each fused training launch
            loss = ray_loss(ret, torch.arange(n))
            if hier:
                loss = loss + ray_loss(ret, torch.arange(n), key="rgb0")
            loss.backward()
            # what pytest=True draws (nerf/run.py:375-381, ray.py:117-132,176-184): every draw restarts from seed 0
            def draw(shape, scale=1.0):
                np.random.seed(0)
                return T((np.random.rand(*shape) * scale).astype(np.float32))
            t_rand = draw((n, S)) if jitter else None
            with torch.no_grad():                                                # the depths the passes used, from the inference kernel
                p0 = render.render_pass(rb.to(dev), nc, S, lindisp=lindisp, t_rand=None if t_rand is None else t_rand.to(dev), white_bkgd=white,
                                        noise=draw((n, S), noise_std).to(dev) if noise_std > 0 else None, want=["z_out"],
                                        n_importance=Ni, u=draw((n, Ni)).to(dev) if (hier and jitter) else None)
            what = f"case {case} views={views} n={n} S={S}+{Ni} out_ch={oc} white={white} lindisp={lindisp} jitter={jitter} noise={noise_std}"
            z0 = p0["z_out"].cpu()
            if hier:
                coarse_loss = lambda r, idx: ((r["rgb_map"] - tgt[idx].to(r["raw"])) ** 2).sum() / (3 * n)      # img2mse(rgb0), nerf/run.py:695-697
                f0 = flip_aware_check(sd_c, rb, z0, white, coarse_loss, used(nc), what + " coarse net", noise=draw((n, S), noise_std) if noise_std > 0 else None)
                f1 = flip_aware_check(sd_f, rb, p0["z_fine"].cpu(), white, ray_loss, used(nf), what + " fine net", noise=draw((n, S1), noise_std) if noise_std > 0 else None)
            else:
                f0 = flip_aware_check(sd_c, rb, z0, white, ray_loss, used(nc), what, noise=draw((n, S), noise_std) if noise_std > 0 else None)
                f1 = (0, 0)
            worst_flips = max(worst_flips, f0[0], f1[0])
            print(f"\n[parity] {what}: gradients within 2e-5 of float64 (flips/risky coarse {f0}, fine {f1})")
