"""GPU gradient parity: hand-written backward kernels against torch autograd
run through the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

import cases
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(autouse=True)
def _grad():
    with torch.enable_grad():
        yield


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def relclose(a, b, rtol, atol, what):
    a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    lim = atol + rtol * np.abs(b)
    assert np.all(err <= lim), f"{what}: max err {err.max():.3e} (|ref| up to {np.abs(b).max():.3e}), worst excess {(err-lim).max():.3e}"


@pytest.mark.parametrize("S,white,use_noise", [(64, True, False), (192, False, True), (37, True, True), (2, False, False)])
def test_raw2outputs_backward(dev, S, white, use_noise):
    import swnerf.ray as ray
    rng = np.random.default_rng(500 + S)
    N = 41
    raw = (rng.standard_normal((N, S, 4)) * 1.5).astype(np.float32)
    raw[..., 3] = (rng.standard_normal((N, S)) * 3.0 - 0.5).astype(np.float32)
    raw[0, :, 3] = -2.0                                   # empty ray: every sigma gradient is exactly 0
    z = np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1)
    d = rng.standard_normal((N, 3)).astype(np.float32)
    noise = (rng.standard_normal((N, S)) * 0.5).astype(np.float32) if use_noise else None
    gr, gd, ga, gw, gdep = (rng.standard_normal(s).astype(np.float32) for s in ((N, 3), (N,), (N,), (N, S), (N,)))

    def loss_of(outs, mod):
        rgb, disp, acc, w, depth = outs
        ok = ~torch.isnan(disp)
        return ((rgb * mod(gr)).sum() + (torch.where(ok, disp, torch.zeros_like(disp)) * mod(gd) * 0.05).sum()
                + (acc * mod(ga)).sum() + (w * mod(gw)).sum() + (depth * mod(gdep)).sum())

    r_cpu = T(raw).requires_grad_(True)
    loss_of(O.raw2outputs(r_cpu, T(z), T(d), 0., white, noise=None if noise is None else T(noise)), T).backward()
    r_gpu = T(raw).to(dev).requires_grad_(True)
    outs = ray.raw2outputs(r_gpu, T(z).to(dev), T(d).to(dev), 0, white, noise=None if noise is None else T(noise).to(dev))
    assert outs[0].requires_grad
    loss_of(outs, lambda a: T(a).to(dev)).backward()
    relclose(r_gpu.grad, r_cpu.grad, rtol=2e-4, atol=2e-6, what=f"d raw (S={S})")
    assert float(r_gpu.grad[0, :, 3].abs().max()) == 0.0
    # only d(rgb_map): the loss of nerf/run.py:688-697
    r_cpu.grad = None
    O.raw2outputs(r_cpu, T(z), T(d), 0., white)[0].pow(2).sum().backward()
    r_gpu.grad = None
    ray.raw2outputs(r_gpu, T(z).to(dev), T(d).to(dev), 0, white)[0].pow(2).sum().backward()
    relclose(r_gpu.grad, r_cpu.grad, rtol=2e-4, atol=2e-6, what="d raw from rgb only")


def _static_net(dev, sd_np):
    import swnerf.model as model
    m = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: T(v) for k, v in sd_np.items()})
    return m.to(dev)


def _grad_check(ours, ref, what, rtol=2e-4):
    """per tensor: max |delta| <= rtol * max |ref grad| (gradients are sums over rows of fp32 products)"""
    for name, g in ours.items():
        r = ref[name].double().numpy()
        d = np.abs(g.detach().cpu().double().numpy() - r).max()
        scale = max(np.abs(r).max(), 1e-12)
        assert d <= rtol * scale, f"{what} {name}: max err {d:.3e} vs max |grad| {scale:.3e} (ratio {d/scale:.2e})"


@pytest.mark.parametrize("M", [300, 32, 1])
def test_mlp_backward_matches_autograd(dev, M):
    sd_np, _ = cases.weights_static()
    g = cases.g4_inputs()
    x = T(g["x"][:M])
    G = T(np.random.default_rng(9).standard_normal((M, 4)).astype(np.float32))
    sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(sd_np).items()}
    (O.nerf_mlp(sd, x) * G).sum().backward()
    net = _static_net(dev, sd_np)
    out = net(x.to(dev))
    assert out.requires_grad and out.shape == (M, 4)
    (out * G.to(dev)).sum().backward()
    _grad_check({k: p.grad for k, p in net.named_parameters()}, {k: v.grad for k, v in sd.items()}, f"M={M}")
    # a second backward accumulates (what an optimizer loop without zero_grad would see)
    (net(x.to(dev)) * G.to(dev)).sum().backward()
    _grad_check({k: p.grad / 2 for k, p in net.named_parameters()}, {k: v.grad for k, v in sd.items()}, "accumulated")


def test_training_step_matches_autograd(dev):
    """One step of the reference's training loss (nerf/run.py:688-697) on the differentiable op path:
    coarse-only and coarse+fine, gradients of every parameter vs autograd through the CPU oracle."""
    import swnerf.embedder as embedder, swnerf.render as render
    sd_c, sd_f = cases.weights_static()
    g = cases.g7_inputs(n=48, seed=77)
    rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6.)
    target = T(np.random.default_rng(3).uniform(0, 1, (48, 3)).astype(np.float32))
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = embedder.get_embedder(4, 3, 0)
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=1024)   # several chunks
    img2mse = lambda x, y: torch.mean((x - y) ** 2)
    # ---- coarse only
    oc = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(sd_c).items()}
    r = O.render_rays(rb, oc, None, 64, 0, white_bkgd=True)
    img2mse(r["rgb_map"], target).backward()
    net_c = _static_net(dev, sd_c)
    ret = render.render_rays(rb.to(dev), net_c, q, 64, N_importance=0, white_bkgd=True, perturb=0., raw_noise_std=0.)
    loss = img2mse(ret["rgb_map"], target.to(dev))
    loss.backward()
    _grad_check({k: p.grad for k, p in net_c.named_parameters()}, {k: v.grad for k, v in oc.items()}, "coarse-only step", rtol=5e-4)
    from flipcheck import flip_aware_check                # float64 truth + exact ReLU-flip accounting: 2e-5 (tests/flipcheck.py)
    mse48 = lambda r, idx: ((r["rgb_map"] - target[idx].to(r["raw"])) ** 2).sum() / (3 * 48)
    fl = flip_aware_check(sd_c, rb, O.coarse_z(rb[:, 6:7], rb[:, 7:8], 64), True, mse48, {k: p.grad for k, p in net_c.named_parameters()}, "coarse-only step vs float64")
    print(f"\n[parity] coarse-only training step (48 x 64 rows): every gradient within 2e-5 of float64; ReLU flips (of risky units) {fl}")
    # ---- coarse + fine, both losses; the resampled depths are detached (nerf/run.py:398) - feed the oracle OUR z_fine
    # so that the comparison is not about sample_pdf's conditioning
    net_c.zero_grad()
    net_f = _static_net(dev, sd_f)
    ret = render.render_rays(rb.to(dev), net_c, q, 64, N_importance=128, network_fine=net_f, white_bkgd=True)
    loss = img2mse(ret["rgb_map"], target.to(dev)) + img2mse(ret["rgb0"], target.to(dev))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(net_c.parameters()) + list(net_f.parameters()))
    # fine net alone on fixed depths: gradient parity
    with torch.no_grad():
        z_fine = render.render_pass(rb.to(dev), net_c, 64, white_bkgd=True, want=[], n_importance=128)["z_fine"].cpu()
    of = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(sd_f).items()}
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z_fine[..., None]
    raw = O.run_network(of, pts, rb[:, -3:])
    img2mse(O.raw2outputs(raw, z_fine, rb[:, 3:6], 0., True)[0], target).backward()
    net_f.zero_grad()
    import swnerf.ray as ray
    raw_g = q(pts.to(dev), rb[:, -3:].to(dev), net_f)
    img2mse(ray.raw2outputs(raw_g, z_fine.to(dev), rb[:, 3:6].to(dev), 0, True)[0], target.to(dev)).backward()
    _grad_check({k: p.grad for k, p in net_f.named_parameters()}, {k: v.grad for k, v in of.items()}, "fine net, fixed depths", rtol=5e-4)
    fl = flip_aware_check(sd_f, rb, z_fine, True, mse48, {k: p.grad for k, p in net_f.named_parameters()}, "fine net on the op path, fixed depths, vs float64")
    print(f"\n[parity] op-path fine net (48 x 192 rows): every gradient within 2e-5 of float64; ReLU flips (of risky units) {fl}")
    # an Adam step on these gradients changes the render (weights are re-packed automatically)
    opt = torch.optim.Adam(list(net_f.parameters()), lr=5e-4, betas=(0.9, 0.999))
    with torch.no_grad():
        before = q(pts.to(dev)[:4], rb[:4, -3:].to(dev), net_f).clone()
    opt.step()
    with torch.no_grad():
        after = q(pts.to(dev)[:4], rb[:4, -3:].to(dev), net_f)
    assert float((after - before).abs().max()) > 1e-5


def test_fused_training_pass_matches_op_path_and_autograd(dev, monkeypatch):
    """The fused pass under autograd (swnerf_render_pass_train / _backward: sampling, encoding, MLP with saved
    activations, compositing and its backward, dX chain - one wave per ray, nothing but act/grad in HBM) against
    (a) the differentiable op path (embed -> cat -> mlp_forward_train -> raw2outputs ...) on the same inputs, every
    output and every parameter gradient, and (b) torch autograd through the CPU oracle on fixed depths.
    Ragged shapes: S not a multiple of 32 (padded tile rows must contribute nothing), N not a multiple of 4."""
    import swnerf.embedder as embedder, swnerf.render as render
    sd_c, sd_f = cases.weights_static()
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = embedder.get_embedder(4, 3, 0)
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    rng = np.random.default_rng(5)
    for n, S, Ni, kw in ((48, 64, 128, dict(white_bkgd=True)),
                         (37, 40, 24, dict(white_bkgd=False, lindisp=True, perturb=1., pytest=True, raw_noise_std=1.0)),
                         (5, 33, 0, dict(white_bkgd=True)),
                         (3, 2, 0, dict(white_bkgd=False)),                      # the smallest pass the reference allows
                         (2, 256, 0, dict(white_bkgd=True, raw_noise_std=0.5, pytest=True)),   # the largest the fused backward holds in LDS
                         (1, 64, 128, dict(white_bkgd=True, retraw=True)),       # one ray; raw returned (nerf/run.py:685)
                         (9, 64, 64, dict(white_bkgd=True, one_net=True))):      # network_fine=None: ONE net for both passes (nerf/run.py:402)
        g = cases.g7_inputs(n=n, seed=90 + n)
        rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6.).to(dev)
        tgt = T(rng.uniform(0, 1, (n, 3)).astype(np.float32)).to(dev)
        wd, wa = T(rng.standard_normal(n).astype(np.float32)).to(dev), T(rng.standard_normal(n).astype(np.float32)).to(dev)

        # the backward runs per chunk of rays (render.TRAIN_BWD_CHUNK_ROWS): force several ragged chunks here
        monkeypatch.setattr(render, "TRAIN_BWD_CHUNK_ROWS", 2048 if n != 48 else 393216)

        def run(op_path):
            if op_path:
                monkeypatch.setenv("SWNERF_TRAIN_OP_PATH", "1")
            else:
                monkeypatch.delenv("SWNERF_TRAIN_OP_PATH", raising=False)
            nc, nf = _static_net(dev, sd_c), _static_net(dev, sd_f)
            kw2 = {k: v for k, v in kw.items() if k != "one_net"}
            ret = render.render_rays(rb, nc, q, S, N_importance=Ni, network_fine=nf if (Ni and not kw.get("one_net")) else None, **kw2)
            ok = ~torch.isnan(ret["disp_map"])
            # the reference's loss (nerf/run.py:688-697) plus terms that put gradients on disp_map and acc_map too
            loss = torch.mean((ret["rgb_map"] - tgt) ** 2) + 0.01 * (torch.where(ok, ret["disp_map"], torch.zeros_like(wd)) * wd).mean() \
                + 0.1 * (ret["acc_map"] * wa).mean()
            if Ni:
                loss = loss + torch.mean((ret["rgb0"] - tgt) ** 2)
            loss.backward()
            return ret, {("c." + k): p.grad for k, p in nc.named_parameters()} | (
                {("f." + k): p.grad for k, p in nf.named_parameters()} if (Ni and not kw.get("one_net")) else {})

        ret_f, g_f = run(False)
        ret_o, g_o = run(True)
        assert list(ret_f.keys()) == list(ret_o.keys())
        for k in ret_f:                                  # (disp is NaN for empty rays on both paths)
            relclose(ret_f[k].nan_to_num(7.0), ret_o[k].nan_to_num(7.0), rtol=1e-6, atol=2e-7, what=f"fused vs op path {k} (S={S})")
        assert all(v is not None for v in g_f.values())
        worst = 0.0
        for k in g_f:
            r = g_o[k].double().cpu().numpy()
            d = np.abs(g_f[k].double().cpu().numpy() - r).max()
            scale = max(np.abs(r).max(), 1e-12)
            worst = max(worst, d / scale)
            # same arithmetic per element; the split-K atomics of the TN GEMMs add in a different order run to run
            assert d <= 2e-6 * scale, f"S={S} {k}: fused vs op path {d:.3e} of {scale:.3e}"
        print(f"\n[parity] fused training pass vs op path, N={n} S={S}+{Ni}: worst parameter-gradient difference {worst:.2e} of its max")
    # (b) autograd through the oracle: the fine pass on fixed depths, gradients of every parameter
    g = cases.g7_inputs(n=48, seed=77)
    rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6.)
    target = T(np.random.default_rng(3).uniform(0, 1, (48, 3)).astype(np.float32))
    net_c, net_f = _static_net(dev, sd_c), _static_net(dev, sd_f)
    with torch.no_grad():
        z_fine = render.render_pass(rb.to(dev), net_c, 64, white_bkgd=True, want=[], n_importance=128)["z_fine"]
    out = render.render_pass_train(rb.to(dev), net_f, 192, z_vals=z_fine, white_bkgd=True)
    torch.mean((out["rgb_map"] - target.to(dev)) ** 2).backward()
    of = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(sd_f).items()}
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z_fine.cpu()[..., None]
    torch.mean((O.raw2outputs(O.run_network(of, pts, rb[:, -3:]), z_fine.cpu(), rb[:, 3:6], 0., True)[0] - target) ** 2).backward()
    _grad_check({k: p.grad for k, p in net_f.named_parameters()}, {k: v.grad for k, v in of.items()}, "fused fine pass vs autograd", rtol=5e-4)
    # ... and to 2e-5 of each tensor's max against the float64 evaluation, with the ReLU flips of near-zero units accounted
    # for exactly (tests/flipcheck.py): what is left of the 5e-4 above is conditioning of the fp32 oracle, not arithmetic
    from flipcheck import flip_aware_check
    fl = flip_aware_check(sd_f, rb, z_fine.cpu(), True, lambda r, idx: ((r["rgb_map"] - target[idx].to(r["raw"])) ** 2).sum() / (3 * 48),
                          {k: p.grad for k, p in net_f.named_parameters()}, "fused fine pass vs float64 autograd")
    print(f"\n[parity] fused training fine pass (48 x 192 rows, view directions): every gradient within 2e-5 of float64; ReLU flips (of risky units) {fl}")
    # what falls back to the op path still trains: raw requested, or more samples than the fused backward holds in LDS
    ret = render.render_rays(rb.to(dev)[:8], net_c, q, 64, retraw=True, N_importance=0, white_bkgd=True)
    assert ret["raw"].requires_grad
    ret = render.render_rays(rb.to(dev)[:8], net_c, q, 300, N_importance=0, white_bkgd=True)
    ret["rgb_map"].sum().backward()


@pytest.mark.gpu
def test_fused_training_pass_with_unused_outputs(dev, monkeypatch):
    """A loss that uses ONE output of the fused pass: the gradients of the others arrive as None (set_materialize_grads(False) - no
    zero fills, NULL pointers into the backward kernel, csrc/train_kernels.hip).  acc_map alone (no gradient on rgb_map at all) and
    disp_map alone, fused against the op path."""
    import swnerf.embedder as embedder, swnerf.render as render
    sd_c, _ = cases.weights_static()
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = embedder.get_embedder(4, 3, 0)
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    g = cases.g7_inputs(n=21, seed=12)
    rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6.).to(dev)
    wa = T(np.random.default_rng(8).standard_normal(21).astype(np.float32)).to(dev)
    for key in ("acc_map", "disp_map"):
        grads = []
        for op_path in (False, True):
            if op_path:
                monkeypatch.setenv("SWNERF_TRAIN_OP_PATH", "1")
            else:
                monkeypatch.delenv("SWNERF_TRAIN_OP_PATH", raising=False)
            net = _static_net(dev, sd_c)
            ret = render.render_rays(rb, net, q, 48, N_importance=0, white_bkgd=True)
            (ret[key].nan_to_num(0.0) * wa).mean().backward()
            grads.append({k: p.grad for k, p in net.named_parameters()})
        for k in grads[0]:
            assert grads[0][k] is not None, k
            r = grads[1][k].double().cpu().numpy()
            d = np.abs(grads[0][k].double().cpu().numpy() - r).max()
            assert d <= 2e-6 * max(np.abs(r).max(), 1e-12), f"{key} only, {k}: fused vs op path {d:.3e}"


def _dnerf_net(dev, sd_np):
    import swnerf.embedder as embedder, swnerf.model as model
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    m = model.DirectTemporalNeRF(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, output_ch=5, skips=[4],
                                 use_viewdirs=True, embed_fn=embed_fn, zero_canonical=True)
    m.load_state_dict({k: T(v) for k, v in sd_np.items()})
    return m.to(dev)


@pytest.mark.parametrize("M", [300, 32, 1])
def test_dnerf_mlp_backward_matches_autograd(dev, M):
    """DirectTemporalNeRF.forward at t != 0 with gradients on BOTH outputs (out and dx = position_delta, the TV-loss
    operand of d_nerf/run_dnerf.py:690-725): every parameter gradient of `_time`, `_time_out` and `_occ`."""
    sd_np = cases.weights_dnerf()
    x = T(cases.g4_inputs()["x"][:M])
    t_emb = O.embed(torch.full((M, 1), 0.5), 10)
    rng = np.random.default_rng(19)
    G, Gdx = T(rng.standard_normal((M, 4)).astype(np.float32)), T(rng.standard_normal((M, 3)).astype(np.float32))
    net = _dnerf_net(dev, sd_np)
    out, dx = net(x.to(dev), [t_emb.to(dev), t_emb.to(dev)])
    assert out.requires_grad and dx.requires_grad and out.shape == (M, 4) and dx.shape == (M, 3)
    ((out * G.to(dev)).sum() + (dx * Gdx.to(dev)).sum()).backward()
    # forward values: the same as the inference kernel's
    with torch.no_grad():
        out_i, dx_i = net(x.to(dev), [t_emb.to(dev), t_emb.to(dev)])
    relclose(dx, dx_i, rtol=1e-5, atol=1e-6, what="dx train vs inference")
    sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(sd_np).items()}
    o_ref, dx_ref = O.dnerf_mlp(sd, x, t_emb, dx_value=dx.detach().cpu())
    relclose(dx, dx_ref, rtol=1e-5, atol=2e-6, what="dx")
    relclose(out, o_ref, rtol=2e-4, atol=2e-5, what="out at equal dx")
    ((o_ref * G).sum() + (dx_ref * Gdx).sum()).backward()
    _grad_check({k: p.grad for k, p in net.named_parameters()}, {k: v.grad for k, v in sd.items()}, f"dnerf M={M}", rtol=3e-4)
    # only the TV-style gradient (d out = None inside autograd): `_occ` gets exactly zero, `_time` the direct path
    net.zero_grad()
    out, dx = net(x.to(dev), [t_emb.to(dev), t_emb.to(dev)])
    (dx * Gdx.to(dev)).sum().backward()
    for v in sd.values():
        v.grad = None
    (O.dnerf_mlp(sd, x, t_emb)[1] * Gdx).sum().backward()
    ours = {k: p.grad for k, p in net.named_parameters() if k.startswith("_time")}
    _grad_check(ours, {k: sd[k].grad for k in ours}, "dx-only", rtol=3e-4)
    assert all(float(p.grad.abs().max()) == 0.0 for k, p in net.named_parameters() if k.startswith("_occ"))


def test_dnerf_zero_time_trains_canonical_only(dev):
    """t == 0 with zero_canonical (model.py:143-145): dx = 0, only `_occ` receives gradients."""
    sd_np = cases.weights_dnerf()
    M = 100
    x = T(cases.g4_inputs()["x"][:M])
    t_emb = O.embed(torch.zeros((M, 1)), 10)
    G = T(np.random.default_rng(23).standard_normal((M, 4)).astype(np.float32))
    net = _dnerf_net(dev, sd_np)
    out, dx = net(x.to(dev), [t_emb.to(dev), t_emb.to(dev)])
    assert float(dx.abs().max()) == 0.0
    (out * G.to(dev)).sum().backward()
    sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(sd_np).items()}
    (O.dnerf_mlp(sd, x, t_emb)[0] * G).sum().backward()
    ours = {k: p.grad for k, p in net.named_parameters() if k.startswith("_occ")}
    _grad_check(ours, {k: sd[k].grad for k in ours}, "t=0")
    assert all(p.grad is None for k, p in net.named_parameters() if k.startswith("_time"))


def test_dnerf_training_step_with_tv_loss(dev):
    """The D-NeRF training loss (d_nerf/run_dnerf.py:690-725): image loss at the frame's time plus the TV term between
    position_delta at two times on the SAME depths (`z_vals=extras['z_vals'].detach()`), through render_rays on the
    differentiable op path.  Compared with autograd through the CPU oracle; the tolerance is set by gamma()'s top
    band, which turns the oracle's 1e-7 difference in dx into ~5e-5 in the re-embedded features."""
    import swnerf.embedder as embedder, swnerf.render_dnerf as rd
    sd_np = cases.weights_dnerf()
    g = cases.g8_inputs(n=40)
    rb = lambda t: O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6., frame_time=t)
    target = T(np.random.default_rng(5).uniform(0, 1, (40, 3)).astype(np.float32))
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = embedder.get_embedder(4, 3, 0)
    embedtime_fn, _ = embedder.get_embedder(10, 1, 0)
    q = lambda inputs, viewdirs, ts, network_fn: rd.run_network(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn, netchunk=1024)
    img2mse = lambda x, y: torch.mean((x - y) ** 2)
    tv_w = 0.1

    sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(sd_np).items()}
    r1 = O.render_rays_dnerf(rb(0.5), sd, 64, white_bkgd=True)
    r0 = O.render_rays_dnerf(rb(0.45), sd, 64, white_bkgd=True)
    (img2mse(r1["rgb_map"], target) + tv_w * (r1["position_delta"] - r0["position_delta"]).pow(2).sum()).backward()

    net = _dnerf_net(dev, sd_np)
    e1 = rd.render_rays(rb(0.5).to(dev), net, q, 64, retraw=True, white_bkgd=True, perturb=0., raw_noise_std=0.)
    e0 = rd.render_rays(rb(0.45).to(dev), net, q, 64, retraw=True, white_bkgd=True, z_vals=e1["z_vals"].detach())
    assert e1["position_delta"].requires_grad and e1["rgb_map"].requires_grad
    loss = img2mse(e1["rgb_map"], target.to(dev)) + tv_w * (e1["position_delta"] - e0["position_delta"]).pow(2).sum()
    loss.backward()
    relclose(e1["position_delta"], r1["position_delta"], rtol=1e-5, atol=2e-6, what="position_delta")
    _grad_check({k: p.grad for k, p in net.named_parameters()}, {k: v.grad for k, v in sd.items()}, "dnerf step", rtol=5e-3)
    # the default D-NeRF configuration (64+128, one net): runs, every parameter gets a finite gradient, Adam moves the render
    net.zero_grad()
    e = rd.render_rays(rb(0.5).to(dev), net, q, 64, N_importance=128, retraw=True, white_bkgd=True)
    (img2mse(e["rgb_map"], target.to(dev)) + tv_w * e["position_delta"].pow(2).sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0 for p in net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    opt.step()
    with torch.no_grad():
        after = rd.render_rays(rb(0.5).to(dev), net, q, 64, N_importance=128, white_bkgd=True)["rgb_map"]
    assert float((after - e["rgb_map"].detach()).abs().max()) > 1e-6


def test_reference_train_call_takes_the_fused_kernels(dev, monkeypatch):
    """The call the reference's train() makes - render(..., retraw=True, **render_kwargs_train) with perturb=1
    (nerf/run.py:684-686, d_nerf/run_dnerf.py:686-688) - must run on the FUSED training kernels, not quietly on the op
    path: count the C entry points that get called."""
    import swnerf.embedder as embedder, swnerf.render as render, swnerf.render_dnerf as rd
    from swnerf import _lib
    L = _lib.lib()
    calls = {}

    class Spy:
        def __init__(self, lib):
            object.__setattr__(self, "_l", lib)

        def __getattr__(self, name):
            f = getattr(self._l, name)
            if not name.startswith("swnerf_"):
                return f

            def wrapped(*a, **k):
                calls[name] = calls.get(name, 0) + 1
                return f(*a, **k)
            return wrapped
    spy = Spy(L)
    monkeypatch.setattr(_lib, "lib", lambda: spy)
    monkeypatch.delenv("SWNERF_TRAIN_OP_PATH", raising=False)
    sd_c, sd_f = cases.weights_static()
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = embedder.get_embedder(4, 3, 0)
    embedtime_fn, _ = embedder.get_embedder(10, 1, 0)
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    g = cases.g7_inputs(n=32, seed=3)
    K, _ = cases.synth.lego_camera(400, 400)
    rays = (T(g["rays_o"]).to(dev), T(g["rays_d"]).to(dev))
    nc, nf = _static_net(dev, sd_c), _static_net(dev, sd_f)
    rgb, disp, acc, extras = render.render(400, 400, K, chunk=1024 * 32, rays=rays, retraw=True, ndc=False, near=2., far=6., use_viewdirs=True,
                                           network_fn=nc, network_query_fn=q, N_samples=64, N_importance=128, network_fine=nf,
                                           white_bkgd=True, perturb=1., raw_noise_std=0.)
    assert sorted(extras.keys()) == ["acc0", "disp0", "raw", "rgb0", "z_std"] and extras["raw"].shape == (32, 192, 4)
    (rgb.pow(2).mean() + extras["rgb0"].pow(2).mean()).backward()
    assert calls.get("swnerf_render_pass_train") == 2 and calls.get("swnerf_render_pass_backward") == 2
    assert not any(k in calls for k in ("swnerf_mlp_forward_train", "swnerf_mlp_backward_dx", "swnerf_embed", "swnerf_raw2outputs"))
    assert all(p.grad is not None and float(p.grad.abs().max()) > 0 for p in list(nc.parameters()) + list(nf.parameters()))
    # D-NeRF, the shipped one-model configuration
    calls.clear()
    qd = lambda inputs, viewdirs, ts, network_fn: rd.run_network(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn,
                                                                 embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn, netchunk=1024 * 64)
    dn = _dnerf_net(dev, cases.weights_dnerf())
    rgb, disp, acc, extras = rd.render(400, 400, float(K[0][0]), chunk=1024 * 32, rays=rays, frame_time=0.5, retraw=True, ndc=False, near=2.,
                                       far=6., use_viewdirs=True, network_fn=dn, network_query_fn=qd, N_samples=64, N_importance=128,
                                       network_fine=None, white_bkgd=True, perturb=1., raw_noise_std=0., use_two_models_for_fine=False)
    assert extras["position_delta"].shape == (32, 192, 3) and extras["position_delta"].requires_grad
    (rgb.pow(2).mean() + extras["position_delta"].pow(2).sum()).backward()
    assert calls.get("swnerf_render_pass") == 1                       # the no_grad coarse pass that feeds the resampling
    assert calls.get("swnerf_render_pass_train_dnerf") == 1 and calls.get("swnerf_render_pass_backward_dnerf") == 1
    assert not any(k in calls for k in ("swnerf_deform_forward_train", "swnerf_mlp_forward_train", "swnerf_embed", "swnerf_raw2outputs"))
    assert all(p.grad is not None and float(p.grad.abs().max()) > 0 for p in dn.parameters())


def test_fused_dnerf_training_pass_matches_op_path(dev, monkeypatch):
    """The fused D-NeRF training pass (swnerf_render_pass_train_dnerf / _backward_dnerf: deformation net -> x+dx ->
    canonical net -> compositing, and back, one ring over both transposed streams) against the differentiable op path
    on the same inputs: outputs, position_delta and every parameter gradient of `_occ`, `_time`, `_time_out`; with
    gradients on the image, on position_delta (TV term on shared depths) and on the returned raw; ragged S, several
    backward chunks; and t == 0 (zero_canonical: the static fused pass on `_occ`, position_delta = 0)."""
    import swnerf.embedder as embedder, swnerf.render_dnerf as rd, swnerf.render as render
    sd_np = cases.weights_dnerf()
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = embedder.get_embedder(4, 3, 0)
    embedtime_fn, _ = embedder.get_embedder(10, 1, 0)
    q = lambda inputs, viewdirs, ts, network_fn: rd.run_network(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn, netchunk=1024 * 64)
    rng = np.random.default_rng(8)
    for n, S, Ni, tv in ((40, 64, 128, 0.5), (21, 40, 0, 0.25), (12, 64, 128, 0.0), (10, 64, 64, -0.5)):     # tv < 0: two models, t = |tv|
        g = cases.g8_inputs(n=n)
        rb = lambda t: O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6., frame_time=t).to(dev)
        tgt = T(rng.uniform(0, 1, (n, 3)).astype(np.float32)).to(dev)
        wr = T(rng.standard_normal((n, S + Ni, 4)).astype(np.float32)).to(dev) * 1e-3
        monkeypatch.setattr(render, "TRAIN_BWD_CHUNK_ROWS", 4096 if n != 40 else 393216)

        two, tv = tv < 0, abs(tv)

        def run(op_path):
            if op_path:
                monkeypatch.setenv("SWNERF_TRAIN_OP_PATH", "1")
            else:
                monkeypatch.delenv("SWNERF_TRAIN_OP_PATH", raising=False)
            net = _dnerf_net(dev, sd_np)
            fine = _dnerf_net(dev, sd_np) if two else None            # use_two_models_for_fine (run_dnerf.py:410-416)
            kw2 = dict(network_fine=fine, use_two_models_for_fine=two)
            e1 = rd.render_rays(rb(tv), net, q, S, N_importance=Ni, retraw=True, white_bkgd=True, perturb=0., raw_noise_std=0., **kw2)
            e0 = rd.render_rays(rb(tv + 0.03), net, q, S, N_importance=Ni, retraw=True, white_bkgd=True, z_vals=e1["z_vals"].detach(), **kw2)
            loss = torch.mean((e1["rgb_map"] - tgt) ** 2) + 0.1 * (e1["position_delta"] - e0["position_delta"]).pow(2).sum() / n \
                + (e1["raw"] * wr).sum() + 0.05 * e1["acc_map"].mean()
            if two:
                loss = loss + torch.mean((e1["rgb0"] - tgt) ** 2) + 0.1 * e1["position_delta_0"].pow(2).sum() / n
            loss.backward()
            g_ = {k: p.grad for k, p in net.named_parameters()}
            if two:
                g_.update({"fine." + k: p.grad for k, p in fine.named_parameters()})
            return e1, e0, g_

        f1, f0, g_f = run(False)
        o1, o0, g_o = run(True)
        assert list(f1.keys()) == list(o1.keys()) and list(f0.keys()) == list(o0.keys())
        for k in f1:
            relclose(f1[k].nan_to_num(7.0), o1[k].nan_to_num(7.0), rtol=2e-6, atol=5e-7, what=f"fused vs op path {k} (t={tv})")
        relclose(f0["position_delta"], o0["position_delta"], rtol=2e-6, atol=5e-7, what="position_delta at the second time")
        if tv == 0.0:
            assert float(f1["position_delta"].abs().max()) == 0.0
        worst = 0.0
        for k in g_o:
            if g_o[k] is None:
                assert g_f[k] is None or float(g_f[k].abs().max()) == 0.0, k
                continue
            r = g_o[k].double().cpu().numpy()
            d = np.abs(g_f[k].double().cpu().numpy() - r).max()
            scale = max(np.abs(r).max(), 1e-12)
            worst = max(worst, d / scale)
            assert d <= 5e-6 * scale, f"t={tv} {k}: fused vs op path {d:.3e} of {scale:.3e}"
        print(f"\n[parity] fused D-NeRF training pass vs op path, N={n} S={S}+{Ni} t={tv}: worst parameter-gradient difference {worst:.2e} of its max")


@pytest.mark.parametrize("M", [4096, 5003, 131072 + 17])
def test_gemm_tn_256x256_dma_path(dev, M):
    """The 256x256 weight-gradient GEMM (LDS-DMA double-buffered kernel, taken for M >= 4096) against torch in float64:
    ragged row counts (last slab partial, last workgroup short), operands that are column windows of wider buffers
    (lda = ldb = 2432 like `grad` / `act`), accumulation into a non-zero C, and the bias column sums."""
    from swnerf import _lib
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M)
    A = torch.randn((M, 2432), generator=g).to(dev)
    B = torch.randn((M, 2432), generator=g).to(dev)
    C0 = torch.randn((256, 319), generator=g).to(dev)
    C, bias = C0.clone(), torch.zeros(256, device=dev)
    st = _lib.stream_of(A)
    _lib.check(L.swnerf_gemm_tn(A.data_ptr() + 4 * 512, A.stride(0), 256, B.data_ptr() + 4 * 1024, B.stride(0), 256, M,
                                C.data_ptr() + 4 * 63, C.stride(0), _lib.ptr(bias), st), "gemm_tn")
    ref = A[:, 512:768].double().T @ B[:, 1024:1280].double()
    scale = float(ref.abs().max())
    assert float((C[:, 63:] - C0[:, 63:] - ref.float()).abs().max()) <= 2e-5 * scale
    assert torch.equal(C[:, :63], C0[:, :63])
    bref = A[:, 512:768].double().sum(0)
    assert float((bias - bref.float()).abs().max()) <= 2e-5 * float(bref.abs().max())


@pytest.mark.parametrize("No,Ni,a_col,b_col,lda,ldb", [
    (256, 64, 0, 0, 2432, 96),        # pts_linears.0 / _time.0: d pre_0 x the gamma(x) slots of xs
    (128, 256, 2304, 2048, 2432, 2432),   # views_linears.0: d pre_views x feature
    (128, 32, 2304, 64, 2432, 96),    # views_linears.0: x the gamma(d) slots
    (256, 32, 0, 64, 2432, 96),       # _time.0: x the gamma(t) slots
    (4, 128, 0, 2304, 4, 2432),       # rgb_linear in its 4-row form: d raw x the view hidden layer
    (4, 256, 0, 1792, 4, 2432),       # _time_out in its 4-row form
    (8, 100, 4, 8, 16, 128),          # not a shape of the step: widths that are no multiples of 32
    (128, 283, 0, 0, 128, 283),       # views_linears.0 of a W = 256 net on the GENERIC path: <= 128 rows of C, more than 256 columns -
    (64, 300, 0, 0, 64, 300),         # round 2 decoded the grid of this case wrongly (columns past 256 never accumulated, A read out of its rows)
    (256, 319, 0, 0, 256, 319),       # a skip layer on the generic path: both halves of C's rows, two column blocks
])
@pytest.mark.parametrize("M", [4096, 50000 + 7])
def test_gemm_tn_skinny_shapes(dev, M, No, Ni, a_col, b_col, lda, ldb):
    """The skinny weight-gradient GEMMs (double-buffered LDS-DMA kernel with a wave grid per shape, round 3) against torch in
    float64: operands that are column windows of wider buffers, ragged row counts, accumulation into a non-zero window of
    a wider C, bias column sums, and the columns next to the window untouched."""
    from swnerf import _lib
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + 7 * No + Ni)
    A = torch.randn((M, lda), generator=g).to(dev)
    B = torch.randn((M, ldb), generator=g).to(dev)
    C0 = torch.randn((No, Ni + 9), generator=g).to(dev)
    C, bias = C0.clone(), torch.zeros(No, device=dev)
    # C's window must keep 4-byte alignment only (atomics); A and B windows are 16-byte aligned here
    _lib.check(L.swnerf_gemm_tn(A.data_ptr() + 4 * a_col, lda, No, B.data_ptr() + 4 * b_col, ldb, Ni, M,
                                C.data_ptr() + 4 * 5, C.stride(0), _lib.ptr(bias), _lib.stream_of(A)), "gemm_tn")
    ref = A[:, a_col:a_col + No].double().T @ B[:, b_col:b_col + Ni].double()
    scale = float(ref.abs().max())
    assert float((C[:, 5:5 + Ni] - C0[:, 5:5 + Ni] - ref.float()).abs().max()) <= 2e-5 * scale
    assert torch.equal(C[:, :5], C0[:, :5]) and torch.equal(C[:, 5 + Ni:], C0[:, 5 + Ni:])
    bref = A[:, a_col:a_col + No].double().sum(0)
    assert float((bias - bref.float()).abs().max()) <= 2e-5 * float(bref.abs().max())


def test_training_loop_through_create_nerf(dev, tmp_path):
    """The core of train() (nerf/run.py:635-735) on the build: create_nerf -> render(**render_kwargs_train) -> img2mse on
    rgb and rgb0 -> backward -> Adam step -> exponential lr decay -> checkpoint.  The loss on a fixed batch must fall."""
    from types import SimpleNamespace
    import swnerf.render as render, swnerf.runner as runner, swnerf.checkpoint as checkpoint
    args = SimpleNamespace(expname="loop", basedir=str(tmp_path), netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                           lrate=5e-4, lrate_decay=500, netchunk=1024 * 64, no_reload=False, ft_path=None, N_samples=64,
                           N_importance=128, perturb=1., use_viewdirs=True, i_embed=0, multires=10, multires_views=4,
                           raw_noise_std=0., dataset_type="blender", white_bkgd=True, no_ndc=False, lindisp=False)
    torch.manual_seed(0)
    train_kw, test_kw, start, grad_vars, optimizer = runner.create_nerf(args, device=dev)
    g = cases.g7_inputs(n=256, seed=11)
    rays = (T(g["rays_o"]).to(dev), T(g["rays_d"]).to(dev))
    target = T(np.random.default_rng(2).uniform(0, 1, (256, 3)).astype(np.float32)).to(dev)
    K = cases.synth.lego_camera(400, 400)[0]
    img2mse = lambda x, y: torch.mean((x - y) ** 2)
    losses = []
    for i in range(start, start + 25):
        rgb, disp, acc, extras = render.render(400, 400, K, chunk=1024 * 32, rays=rays, near=2., far=6., **train_kw)
        optimizer.zero_grad()
        loss = img2mse(rgb, target) + img2mse(extras['rgb0'], target)
        loss.backward()
        optimizer.step()
        new_lrate = args.lrate * (0.1 ** ((i + 1) / (args.lrate_decay * 1000)))                  # run.py:704-708
        for pg in optimizer.param_groups:
            pg['lr'] = new_lrate
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses
    path = checkpoint.save_checkpoint(args.basedir, args.expname, 25, 25, train_kw['network_fn'], train_kw['network_fine'], optimizer)
    _, test2, start2, _, _ = runner.create_nerf(args, device=dev)
    assert start2 == 25 and path.endswith("000025.tar")
    with torch.no_grad():                                                                         # the reloaded nets render the same
        a = render.render(400, 400, K, chunk=1024 * 32, rays=rays, near=2., far=6., **test_kw)[0]
        b = render.render(400, 400, K, chunk=1024 * 32, rays=rays, near=2., far=6., **test2)[0]
    assert torch.equal(a, b)


@pytest.mark.parametrize("M", [4096, 70000 + 13, 300])
def test_gemm_tn_fused_riders(dev, M):
    """swnerf_gemm_tn_fused against torch in float64: the B2 rider (63 columns of an unaligned [M,90] operand into a
    column window of C), the A2 rider (column 3 of an [M,4] operand, with its bias), both together, and the small-M
    fallback (separate launches)."""
    from swnerf import _lib, model
    L = _lib.lib()
    gen = torch.Generator(device="cpu").manual_seed(7 * M)
    A = torch.randn((M, 2432), generator=gen).to(dev)
    B = torch.randn((M, 2432), generator=gen).to(dev)
    X = torch.randn((M, 90), generator=gen).to(dev)
    D = torch.randn((M, 4), generator=gen).to(dev)
    st = _lib.stream_of(A)
    tol = lambda ref: 2e-5 * float(ref.abs().max())
    Ad, Bd = A[:, 1280:1536].double(), B[:, 1024:1280].double()
    # B2 rider
    C = torch.zeros((256, 319), device=dev)
    bias = torch.zeros(256, device=dev)
    model._gemm_tn_fused(L, st, M, A, 1280, B, 1024, C, 63, bias, B2=X, b2_col=0, Ni2=63, C2=C, c2_col=0)
    ref = torch.cat([Ad.T @ X[:, :63].double(), Ad.T @ Bd], 1)
    assert float((C - ref.float()).abs().max()) <= tol(ref)
    assert float((bias - Ad.sum(0).float()).abs().max()) <= 2e-5 * float(Ad.sum(0).abs().max())
    # A2 rider
    C = torch.zeros((256, 256), device=dev)
    C3, b3 = torch.zeros((1, 256), device=dev), torch.zeros(1, device=dev)
    model._gemm_tn_fused(L, st, M, A, 1280, B, 1024, C, 0, None, A2=D, a2_col=3, No2=1, C3=C3, bias3=b3)
    assert float((C - (Ad.T @ Bd).float()).abs().max()) <= tol(Ad.T @ Bd)
    r3 = D[:, 3:4].double().T @ Bd
    assert float((C3 - r3.float()).abs().max()) <= tol(r3)
    assert abs(float(b3) - float(D[:, 3].double().sum())) <= 2e-5 * max(1.0, abs(float(D[:, 3].double().sum())))
    # both at once (3 rows of A2)
    C = torch.zeros((256, 319), device=dev)
    C3, b3 = torch.zeros((3, 256), device=dev), torch.zeros(3, device=dev)
    model._gemm_tn_fused(L, st, M, A, 1280, B, 1024, C, 63, None, B2=X, b2_col=0, Ni2=63, C2=C, c2_col=0,
                         A2=D, a2_col=0, No2=3, C3=C3, bias3=b3)
    assert float((C - ref.float()).abs().max()) <= tol(ref)
    r3 = D[:, :3].double().T @ Bd
    assert float((C3 - r3.float()).abs().max()) <= tol(r3)
    assert float((b3 - D[:, :3].double().sum(0).float()).abs().max()) <= 2e-5 * float(D[:, :3].double().sum(0).abs().max() + 1)


@pytest.mark.parametrize("M,n_plain", [(4096 + 37, 6), (40000, 1), (3000, 2), (98304, 14)])
def test_gemm_tn_group(dev, M, n_plain):
    """swnerf_gemm_tn_group - the 256 x 256 weight-gradient GEMMs of one row chunk as ONE launch, the workgroups dealt out
    over the items - against torch in float64: plain items, one with the B2 rider (64 slot columns), one with the A2 rider
    (column 3 of a [M,4] operand, with its bias), an item with BOTH riders (split off), ragged M (a last slab of 5 rows,
    slices of unequal length), the small-M fallback, and accumulation into non-zero C (C += ...)."""
    from swnerf import _lib, model
    L = _lib.lib()
    gen = torch.Generator(device="cpu").manual_seed(11 * M + n_plain)
    A = torch.randn((M, 2432), generator=gen).to(dev)
    B = torch.randn((M, 2432), generator=gen).to(dev)
    X = torch.randn((M, 96), generator=gen).to(dev)
    D = torch.randn((M, 4), generator=gen).to(dev)
    tol = lambda ref: 2e-5 * float(ref.abs().max())
    grp = model._Group(_lib.stream_of(A))
    want = []
    for k in range(n_plain):
        a_col, b_col = 256 * (k % 9), 256 * ((k + 3) % 8)
        C, bias = torch.ones((256, 256), device=dev), torch.full((256,), 2.0, device=dev)
        model._gemm_tn(L, grp, M, A, a_col, 256, B, b_col, 256, C, 0, bias if k % 2 == 0 else None)
        Ad, Bd = A[:, a_col:a_col + 256].double(), B[:, b_col:b_col + 256].double()
        want.append((C, 1.0 + Ad.T @ Bd, f"plain {k}"))
        if k % 2 == 0:
            want.append((bias, 2.0 + Ad.sum(0), f"bias {k}"))
    Ad, Bd = A[:, 1280:1536].double(), B[:, 1024:1280].double()
    C5, c5s, b5 = torch.zeros((256, 319), device=dev), torch.zeros((256, 64), device=dev), torch.zeros(256, device=dev)
    model._gemm_tn_fused(L, grp, M, A, 1280, B, 1024, C5, 63, b5, B2=X, b2_col=0, Ni2=64, C2=c5s, c2_col=0)
    want += [(C5[:, 63:], Ad.T @ Bd, "B2 item main"), (c5s, Ad.T @ X[:, :64].double(), "B2 rider"), (b5, Ad.sum(0), "B2 item bias")]
    Cf, C3, b3 = torch.zeros((256, 256), device=dev), torch.zeros((1, 256), device=dev), torch.zeros(1, device=dev)
    model._gemm_tn_fused(L, grp, M, A, 2048, B, 1792, Cf, 0, None, A2=D, a2_col=3, No2=1, C3=C3, bias3=b3)
    Af, Bf = A[:, 2048:2304].double(), B[:, 1792:2048].double()
    want += [(Cf, Af.T @ Bf, "A2 item main"), (C3, D[:, 3:4].double().T @ Bf, "A2 rider"), (b3, D[:, 3].double().sum().reshape(1), "A2 bias")]
    Cb, cbs, C3b, b3b = torch.zeros((256, 256), device=dev), torch.zeros((256, 32), device=dev), torch.zeros((3, 256), device=dev), torch.zeros(3, device=dev)
    model._gemm_tn_fused(L, grp, M, A, 512, B, 256, Cb, 0, None, B2=X, b2_col=64, Ni2=32, C2=cbs, c2_col=0, A2=D, a2_col=0, No2=3, C3=C3b, bias3=b3b)
    Ab, Bb = A[:, 512:768].double(), B[:, 256:512].double()
    want += [(Cb, Ab.T @ Bb, "both riders main"), (cbs, Ab.T @ X[:, 64:96].double(), "both: B2"), (C3b, D[:, :3].double().T @ Bb, "both: A2"),
             (b3b, D[:, :3].double().sum(0), "both: bias3")]
    assert len(grp.items) == n_plain + 3
    grp.launch(L, M)
    assert not grp.items
    for got, ref, what in want:
        assert float((got.double() - ref).abs().max()) <= tol(ref) + 1e-4, f"{what} (M={M}): {float((got.double() - ref).abs().max()):.3e} of {float(ref.abs().max()):.3e}"


def test_feature_finish_kernel(dev):
    """swnerf_feature_finish against the formulas in float64: d views_linears.0.weight[:, :256] += G W_f^T + db_hv (x) b_f,
    d feature_linear.weight += Wv_f^T G, d feature_linear.bias += Wv_f^T db_hv, alpha_linear from row 3 of the 4-row form;
    accumulating into non-zero gradients, Wv with its real leading dimension (283)."""
    from swnerf import _lib
    L = _lib.lib()
    gen = torch.Generator(device="cpu").manual_seed(5)
    r = lambda *s: torch.randn(s, generator=gen).to(dev)
    G, db_hv, Wv, W_f, b_f, a4w, a4b = r(128, 256), r(128), r(128, 283), r(256, 256), r(256), r(4, 256), r(4)
    dWv, dW_f, db_f, dWa, dba = r(128, 283), r(256, 256), r(256), r(1, 256), r(1)
    want = (dWv.double().clone(), dW_f.double() + Wv[:, :256].double().t() @ G.double(), db_f.double() + Wv[:, :256].double().t() @ db_hv.double(),
            dWa.double() + a4w[3:4].double(), dba.double() + a4b[3:4].double())
    want[0][:, :256] += G.double() @ W_f.double().t() + torch.outer(db_hv.double(), b_f.double())
    _lib.check(L.swnerf_feature_finish(_lib.ptr(G), _lib.ptr(db_hv), _lib.ptr(Wv), 283, _lib.ptr(W_f), _lib.ptr(b_f), _lib.ptr(a4w), _lib.ptr(a4b),
                                       _lib.ptr(dWv), 283, _lib.ptr(dW_f), _lib.ptr(db_f), _lib.ptr(dWa), _lib.ptr(dba), _lib.stream_of(G)), "feature_finish")
    for got, ref, what in zip((dWv, dW_f, db_f, dWa, dba), want, ("dWv", "dW_f", "db_f", "dW_alpha", "db_alpha")):
        assert float((got.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), what
    with pytest.raises(RuntimeError):
        _lib.check(L.swnerf_feature_finish(None, _lib.ptr(db_hv), _lib.ptr(Wv), 283, _lib.ptr(W_f), _lib.ptr(b_f), _lib.ptr(a4w), _lib.ptr(a4b),
                                           _lib.ptr(dWv), 283, _lib.ptr(dW_f), _lib.ptr(db_f), _lib.ptr(dWa), _lib.ptr(dba), _lib.stream_of(G)), "feature_finish")


@pytest.mark.parametrize("M", [393216 // 8 + 5, 4099, 7, 16, 300000])
def test_canon_narrow_grads_kernel(dev, M):
    """swnerf_canon_narrow_grads - the five narrow weight-gradient products of the canonical net in one pass over the rows -
    against torch in float64: ragged M (last slab of 5 / 3 / 7 rows, slices of unequal length, fewer rows than a slab),
    accumulation into non-zero outputs, every bias."""
    from swnerf import _lib
    L = _lib.lib()
    gen = torch.Generator(device="cpu").manual_seed(3 * M + 1)
    r = lambda *s: torch.randn(s, generator=gen).to(dev)
    grad, act, xs, d_out = r(M, 2432), r(M, 2432), r(M, 96), r(M, 4)
    outs = dict(c0s=r(256, 64), cvs=r(128, 32), G=r(128, 256), a4w=r(4, 256), rgb4=r(4, 128), b_l0=r(256), b_hv=r(128), a4b=r(4), rgb4b=r(4))
    d = lambda t_: t_.double()
    dp0, dphv, h7, hv = d(grad[:, :256]), d(grad[:, 2304:2432]), d(act[:, 1792:2048]), d(act[:, 2304:2432])
    want = dict(c0s=d(outs["c0s"]) + dp0.T @ d(xs[:, :64]), cvs=d(outs["cvs"]) + dphv.T @ d(xs[:, 64:96]), G=d(outs["G"]) + dphv.T @ h7,
                a4w=d(outs["a4w"]) + d(d_out).T @ h7, rgb4=d(outs["rgb4"]) + d(d_out).T @ hv, b_l0=d(outs["b_l0"]) + dp0.sum(0),
                b_hv=d(outs["b_hv"]) + dphv.sum(0), a4b=d(outs["a4b"]) + d(d_out).sum(0), rgb4b=d(outs["rgb4b"]) + d(d_out).sum(0))
    _lib.check(L.swnerf_canon_narrow_grads(_lib.ptr(grad), 2432, _lib.ptr(act), 2432, _lib.ptr(xs), _lib.ptr(d_out), M,
                                           *[_lib.ptr(outs[k]) for k in ("c0s", "cvs", "G", "a4w", "rgb4", "b_l0", "b_hv", "a4b", "rgb4b")],
                                           _lib.stream_of(grad)), "canon_narrow_grads")
    for k, ref in want.items():
        err = float((d(outs[k]) - ref).abs().max())
        assert err <= 2e-5 * float(ref.abs().max()) + 1e-5, f"{k} (M={M}): {err:.3e} of {float(ref.abs().max()):.3e}"
