"""Shapes the fused kernels are not built for run layer by layer on the generic fp32-MFMA GEMM kernels
(csrc/generic_kernels.hip, swnerf/generic.py): `use_viewdirs=False` - the reference's argparse default, utils.py:26-29,
model.py:59-60 -, other depths / widths / skip sets, DirectTemporalNeRF at D=4.  Against the reference's own outputs
(G11, tests/golden/make_golden_generic.py) and, for gradients, torch autograd through the CPU oracle."""
import numpy as np
import pytest
import torch

import cases
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, equal_nan=True, err_msg=what)


def _net(dev, name):
    import swnerf.model as model
    m = model.vallina_NeRF(**cases.G11_NETS[name])
    m.load_state_dict({k: T(v) for k, v in cases.g11_weights(name).items()}, strict=True)
    return m.to(dev)


def _embedded(dev, kw, g):
    import swnerf.embedder as embedder
    L = (kw["input_ch"] // 3 - 1) // 2
    x = embedder.get_embedder(L, 3, 0)[0](T(g["pts"]).to(dev))
    if kw["input_ch_views"]:
        x = torch.cat([x, embedder.get_embedder(4, 3, 0)[0](T(g["dirs"]).to(dev))], -1)
    return x


def test_linear_kernel_shapes(dev):
    """swnerf_linear / swnerf_gemm_nn / swnerf_relu_mask on ragged shapes (M, N, K not multiples of the 64 x 64 x 32
    tile; K = 63, 319; a strided input) against float64 matmuls."""
    from swnerf import generic
    rng = np.random.default_rng(31)
    for M, K, N, relu in ((1, 63, 256, True), (300, 319, 256, True), (65, 90, 5, False), (1000, 256, 3, False), (129, 7, 130, True), (0, 8, 4, False)):
        lin = torch.nn.Linear(K, N).to(dev)
        x = T(rng.standard_normal((M, K + 3)).astype(np.float32)).to(dev)[:, 1:K + 1]        # a strided view
        with torch.no_grad():
            y = generic.linear(x, lin, relu=relu)
        ref = x.double() @ lin.weight.double().T + lin.bias.double()
        ref = ref.clamp_min(0) if relu else ref
        assert y.shape == (M, N)
        close(y, ref.float(), atol=2e-5, rtol=1e-5, what=f"linear M={M} K={K} N={N}")
    # more than 65536 row blocks (rows sit on grid.x: grid.y stops at 65535 blocks = 4.19 M rows): run_network with netchunk=None
    # on a full frame reaches this
    M = 65536 * 64 + 77
    lin = torch.nn.Linear(8, 4).to(dev)
    x = torch.randn((M, 8), device=dev)
    with torch.no_grad():
        y = generic.linear(x, lin, relu=False)
        for sl in (slice(0, 64), slice(M - 200, M)):
            close(y[sl], (x[sl].double() @ lin.weight.double().T + lin.bias.double()).float(), atol=2e-5, what="linear, M > 65536 x 64")
    del x, y
    # autograd of one layer: dX, dW, db vs torch
    lin = torch.nn.Linear(90, 37).to(dev)
    x = T(rng.standard_normal((211, 90)).astype(np.float32)).to(dev).requires_grad_(True)
    G = T(rng.standard_normal((211, 37)).astype(np.float32)).to(dev)
    (generic.linear(x, lin, relu=True) * G).sum().backward()
    got = [x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    x.grad = None
    lin.zero_grad()
    (torch.relu(torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())) * G.double()).sum().backward()
    for a, b, w in zip(got, (x.grad, lin.weight.grad, lin.bias.grad), ("dX", "dW", "db")):
        close(a, b.float(), atol=2e-5 * float(b.abs().max()), what=w)


def test_generic_mlps_golden(dev, golden):
    import swnerf.embedder as embedder, swnerf.model as model
    ref, g = golden("g11_generic"), cases.g11_inputs()
    for name, kw in cases.G11_NETS.items():
        net = _net(dev, name).eval()
        assert not net._is_fused_arch()
        with torch.no_grad():
            out = net(_embedded(dev, kw, g))
        close(out, ref[f"mlp_{name}"], atol=5e-5, rtol=1e-4, what=f"vallina_NeRF {name}")
        assert net(_embedded(dev, kw, g)[:0]).shape == (0, kw["output_ch"] if not kw["use_viewdirs"] else 4)
    kw = dict(cases.G11_DNERF)
    e10 = embedder.get_embedder(10, 3, 0)[0]
    dn = model.NeRF.get_by_name("direct_temporal", embed_fn=e10, zero_canonical=True, **kw)
    dn.load_state_dict({k: T(v) for k, v in cases.g11_dnerf_weights().items()}, strict=True)
    dn = dn.to(dev).eval()
    x = _embedded(dev, kw, g)
    et = embedder.get_embedder(10, 1, 0)[0]
    for tv in (0.0, 0.5):
        te = et(torch.full((x.shape[0], 1), tv, device=dev))
        with torch.no_grad():
            out, dx = dn(x, [te, te])
        close(dx, ref[f"dnerf_dx_t{int(tv*10)}"], atol=2e-6, what=f"generic dnerf dx t={tv}")
        close(out, ref[f"dnerf_out_t{int(tv*10)}"], atol=2e-3 if tv else 1e-4, rtol=1e-3, what=f"generic dnerf out t={tv}")   # gamma(x+dx): 2^9 band


def test_render_without_viewdirs_golden(dev, golden):
    """render() / render_rays with use_viewdirs=False (nerf/run.py:137-158 builds an 8-column ray batch; model.py:59-60):
    the reference's own end-to-end outputs, and the dict keys / shapes of the drop-in."""
    import swnerf.embedder as embedder, swnerf.render as render
    ref, g = golden("g11_generic"), cases.g11_inputs()
    net = _net(dev, "novd").eval()
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn = None
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    r = g["rays"]
    K, _ = cases.synth.lego_camera(400, 400)
    # round 3: this shape has a FUSED pass of its own (SWNERF_NET_NOVIEW) - the closure and the net qualify for it
    with torch.no_grad():
        assert render.fused_plan(q, [net, None]) == (10, 0, 0) and render.nets_without_views([net, None])
        rgb, disp, acc, extras = render.render(400, 400, K, rays=(T(r["rays_o"]).to(dev), T(r["rays_d"]).to(dev)), ndc=False, near=2., far=6.,
                                               use_viewdirs=False, network_fn=net, network_query_fn=q, N_samples=32, N_importance=32,
                                               network_fine=None, white_bkgd=True, perturb=0., raw_noise_std=0.)
    assert sorted(extras.keys()) == ["acc0", "disp0", "rgb0", "z_std"] and rgb.shape == (64, 3)
    close(extras["rgb0"], ref["rr_rgb0"], atol=2e-5, what="rgb0 (no resampling in front)")
    close(extras["acc0"], ref["rr_acc0"], atol=2e-5, what="acc0")
    d = (rgb.cpu() - T(ref["rr_rgb_map"])).abs()
    assert float((d <= 2e-4).float().mean()) >= 0.9 and float(d.max()) <= 2e-2, f"rgb_map: within 2e-4 {float((d <= 2e-4).float().mean()):.3f}, max {float(d.max()):.2e}"
    close(extras["z_std"], ref["rr_z_std"], atol=2e-3, what="z_std")


def test_fused_pass_without_viewdirs_golden(dev, golden):
    """The fused render pass for use_viewdirs=False (csrc/render_pass.h VIEWS = false: trunk + output_linear as VALU heads,
    8-column rays; model.py:59-60, nerf/run.py:152-157) against the REFERENCE's own render_rays on non-degenerate weights
    (golden G12, tests/golden/make_golden_noview.py): coarse-only with the 5-channel raw, the hierarchical 64+128 case with
    two nets; then against the CPU oracle at ragged sizes, and against the layer-by-layer generic path it replaces."""
    import swnerf.embedder as embedder, swnerf.render as render, swnerf.model as model
    ref, g = golden("g12_noview"), cases.g12_inputs()
    nets, sds = [], []
    for sd_np in cases.g12_weights():
        m = model.vallina_NeRF(**cases.G12_NET)
        m.load_state_dict({k: T(v) for k, v in sd_np.items()}, strict=True)
        nets.append(m.to(dev).eval())
        sds.append(O.to_torch_sd(sd_np))
    fns = [(lambda e, sd=sd: O.generic_mlp(sd, e, 8, [4], 63, 0, False)) for sd in sds]
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    embeddirs_fn = None
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    rb8 = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), g["near"], g["far"])[:, :8].contiguous()
    with torch.no_grad():
        assert render.fused_plan(q, nets) == (10, 0, 0)
        r = render.render_rays(rb8.to(dev), nets[0], q, 64, retraw=True, N_importance=0, white_bkgd=True)
        assert list(r.keys()) == ["rgb_map", "disp_map", "acc_map", "raw"] and r["raw"].shape == (256, 64, 5)
        close(r["raw"][:16], ref["c_raw"], atol=1e-3, rtol=1e-4, what="raw, 5 channels")
        close(r["rgb_map"], ref["c_rgb_map"], atol=2e-5, what="rgb coarse only")
        close(r["acc_map"], ref["c_acc_map"], atol=2e-5, what="acc coarse only")
        close(r["disp_map"], ref["c_disp_map"], atol=2e-5, rtol=1e-4, what="disp coarse only")
        r = render.render_rays(rb8.to(dev), nets[0], q, 64, retraw=True, N_importance=128, network_fine=nets[1], white_bkgd=True)
        assert list(r.keys()) == ["rgb_map", "disp_map", "acc_map", "raw", "rgb0", "disp0", "acc0", "z_std"] and tuple(r["raw"].shape) == tuple(ref["h_raw_shape"])
        close(r["rgb0"], ref["h_rgb0"], atol=2e-5, what="rgb0")
        close(r["acc0"], ref["h_acc0"], atol=2e-5, what="acc0")
        close(r["disp0"], ref["h_disp0"], atol=2e-5, rtol=1e-4, what="disp0")
        close(r["z_std"], ref["h_z_std"], atol=2e-3, what="z_std")
        for k in ("rgb_map", "acc_map"):
            d = (r[k].cpu() - T(ref[f"h_{k}"])).abs()
            db = float(-10 * torch.log10(((r[k].cpu() - T(ref[f"h_{k}"])) ** 2).mean()))
            print(f"\n[parity] no-viewdirs fused 64+128 vs the reference render, {k}: within 2e-4 {float((d <= 2e-4).float().mean()):.4f}, max {float(d.max()):.2e}, PSNR {db:.1f} dB")
            assert float((d <= 2e-4).float().mean()) >= 0.92 and float(d.max()) <= 2e-2 and db >= 70.0
        # ragged sizes (301 rays: a partly filled last workgroup; 40+24 samples) against the CPU oracle, one net for both passes
        g2 = cases.g7_inputs(n=301, seed=58)
        rb2 = O.make_ray_batch(T(g2["rays_o"]), T(g2["rays_d"]), g2["near"], g2["far"])[:, :8].contiguous()
        r = render.render_rays(rb2.to(dev), nets[0], q, 40, retraw=True, N_importance=24, network_fine=None, white_bkgd=False)
        o = O.render_rays_generic(rb2, fns[0], 40, 24, white_bkgd=False)
        assert r["raw"].shape == (301, 64, 5)
        close(r["rgb0"], o["rgb0"], atol=2e-5, what="ragged rgb0")
        d = (r["rgb_map"].cpu() - o["rgb_map"]).abs()
        print(f"\n[parity] no-viewdirs fused 40+24, 301 rays vs oracle: within 2e-4 {float((d <= 2e-4).float().mean()):.4f}, max {float(d.max()):.2e}")
        assert float((d <= 2e-4).float().mean()) >= 0.92 and float(d.max()) <= 2e-2
        # the generic layer-by-layer path (an opaque closure hides the encoder, so no fused plan) gives the same image
        opaque = lambda a, b, c, _q=q: _q(a, b, c)
        assert render.fused_plan(opaque, [nets[0], None]) is None
        a = render.render_rays(rb8.to(dev), nets[0], q, 64, N_importance=0, white_bkgd=True, retraw=True)
        b = render.render_rays(rb8.to(dev), nets[0], opaque, 64, N_importance=0, white_bkgd=True, retraw=True)
        close(a["raw"], b["raw"], atol=2e-4, rtol=1e-4, what="fused vs generic raw")
        close(a["rgb_map"], b["rgb_map"], atol=2e-5, what="fused vs generic rgb")
        # module.forward on embedded rows (run_network's op path, nerf/run.py:73-87) runs the same trunk + heads as one kernel
        # (swnerf_mlp_forward_noview) and equals the layer-by-layer generic path of the same module
        from swnerf.generic import canonical_forward
        xe = embed_fn(torch.from_numpy(cases.g11_inputs()["pts"]).to(dev))
        close(nets[0](xe), canonical_forward(nets[0], xe), atol=2e-5, rtol=1e-5, what="module.forward: fused vs layer by layer")
        assert nets[0](xe.reshape(3, 100, 63)).shape == (3, 100, 5) and nets[0](xe[:33]).shape == (33, 5)
        # 4-channel head, empty batch, and the wrong column count is an error of the C ABI, not a wrong image
        net4 = model.vallina_NeRF(**dict(cases.G12_NET, output_ch=4))
        sd4 = {k: v for k, v in cases.g12_weights()[0].items()}
        sd4["output_linear.weight"], sd4["output_linear.bias"] = sd4["output_linear.weight"][:4], sd4["output_linear.bias"][:4]
        net4.load_state_dict({k: T(v) for k, v in sd4.items()})
        r4 = render.render_rays(rb8.to(dev)[:9], net4.to(dev).eval(), q, 64, retraw=True, white_bkgd=True)
        assert r4["raw"].shape == (9, 64, 4)
        close(r4["raw"], ref["c_raw"][:9, :, :4], atol=1e-3, rtol=1e-4, what="raw, 4 channels")
        close(r4["rgb_map"], ref["c_rgb_map"][:9], atol=2e-5, what="rgb, 4-channel head")
        assert render.render_rays(rb8.to(dev)[:0], nets[0], q, 64, N_importance=16)["rgb_map"].shape == (0, 3)
        with pytest.raises(RuntimeError):
            render.render_pass(torch.zeros((4, 11), device=dev), nets[0], 64)


def test_generic_training_w256_with_viewdirs(dev):
    """A W = 256 net WITH view directions on the generic path (skips other than [4]): views_linears.0 is 128 x 283 - at most
    128 rows of C and more than 256 columns, the weight-gradient shape whose grid round 2 decoded wrongly (the gradient of the
    view-direction columns stayed zero).  Every parameter gradient against torch autograd through the CPU oracle."""
    import swnerf.model as model
    kw = dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[2, 5], use_viewdirs=True)
    sd_np = cases.generic_state_dict(1190, alpha_bias=-0.5, **kw)
    net = model.vallina_NeRF(**kw)
    net.load_state_dict({k: T(v) for k, v in sd_np.items()}, strict=True)
    net = net.to(dev).train()
    assert not net._is_fused_arch()
    g = cases.g11_inputs()
    x = _embedded(dev, kw, g)
    G = T(np.random.default_rng(13).standard_normal((x.shape[0], 4)).astype(np.float32))
    with torch.enable_grad():
        (net(x) * G.to(dev)).sum().backward()
        sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(sd_np).items()}
        (O.generic_mlp(sd, x.cpu(), kw["D"], kw["skips"], kw["input_ch"], kw["input_ch_views"], True) * G).sum().backward()
    for k, p in net.named_parameters():
        rg = sd[k].grad
        assert float(rg.abs().max()) > 0, k
        assert float((p.grad.cpu() - rg).abs().max()) <= 3e-4 * float(rg.abs().max()), f"{k}: {float((p.grad.cpu() - rg).abs().max()):.3e} of {float(rg.abs().max()):.3e}"
    wg = net.views_linears[0].weight.grad
    assert float(wg[:, 256:].abs().max()) > 0            # the view-direction columns


def test_generic_training_matches_autograd(dev):
    """loss.backward() through the generic path: every parameter gradient of a use_viewdirs=False net and of the D=4
    DirectTemporalNeRF (incl. the gradient through gamma(x + dx)) vs torch autograd through the CPU oracle."""
    import swnerf.embedder as embedder, swnerf.model as model
    g = cases.g11_inputs()
    rng = np.random.default_rng(12)
    with torch.enable_grad():
        for name in ("novd", "small"):
            kw = cases.G11_NETS[name]
            net = _net(dev, name).train()
            x = _embedded(dev, kw, g)
            G = T(rng.standard_normal((x.shape[0], 5 if name == "novd" else 4)).astype(np.float32))
            (net(x) * G.to(dev)).sum().backward()
            sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(cases.g11_weights(name)).items()}
            (O.generic_mlp(sd, x.cpu(), kw["D"], kw["skips"], kw["input_ch"], kw["input_ch_views"], kw["use_viewdirs"]) * G).sum().backward()
            for k, p in net.named_parameters():
                if sd[k].grad is None:                         # views_linears of a use_viewdirs=False net: unused (model.py:59-60)
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0
                    continue
                rg = sd[k].grad
                assert float((p.grad.cpu() - rg).abs().max()) <= 3e-4 * max(float(rg.abs().max()), 1e-12), f"{name} {k}"
        kw = dict(cases.G11_DNERF)
        e10 = embedder.get_embedder(10, 3, 0)[0]
        dn = model.NeRF.get_by_name("direct_temporal", embed_fn=e10, zero_canonical=True, **kw)
        dn.load_state_dict({k: T(v) for k, v in cases.g11_dnerf_weights().items()}, strict=True)
        dn = dn.to(dev).train()
        x = _embedded(dev, kw, g)
        te = embedder.get_embedder(10, 1, 0)[0](torch.full((x.shape[0], 1), 0.5, device=dev))
        G, Gdx = T(rng.standard_normal((x.shape[0], 4)).astype(np.float32)), T(rng.standard_normal((x.shape[0], 3)).astype(np.float32))
        out, dx = dn(x, [te, te])
        ((out * G.to(dev)).sum() + (dx * Gdx.to(dev)).sum()).backward()
        sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch_sd(cases.g11_dnerf_weights()).items()}
        o_ref, dx_ref = O.generic_dnerf_mlp(sd, x.cpu(), te.cpu(), kw["D"], kw["skips"], kw["input_ch"], kw["input_ch_views"], True)
        ((o_ref * G).sum() + (dx_ref * Gdx).sum()).backward()
        for k, p in dn.named_parameters():
            rg = sd[k].grad
            # the gradient through gamma(x+dx) carries the 2^9 band: conditioning, not arithmetic
            assert float((p.grad.cpu() - rg).abs().max()) <= 2e-3 * max(float(rg.abs().max()), 1e-12), f"dnerf {k}"


def test_fused_training_without_viewdirs(dev, monkeypatch):
    """loss.backward() of a use_viewdirs=False run (the reference's argparse default; train() nerf/run.py:684-708) on the
    fused pass: swnerf_render_pass_train with SWNERF_NET_NOVIEW + swnerf_render_pass_backward_noview (compositing backward,
    output_linear^T on the VALU, pts_linears.7..1^T on the MFMA ring) + the weight-gradient GEMMs.  Every parameter gradient
    against the float64 evaluation of the oracle, ReLU flips of near-zero units accounted for exactly (tests/flipcheck.py):
    2e-5 of each tensor's max.
    (a) coarse-only passes: 5- and 4-channel heads, ragged sizes (S not a multiple of 32, N not a multiple of 4, S = 2 and
        S = 256), a gradient on every channel of the returned raw (retraw=True) and on disp / acc, raw noise, several backward
        chunks; (b) the layer-by-layer generic path it replaces (SWNERF_TRAIN_OP_PATH=1) through the same check;
    (c) the hierarchical 64+128 step with two nets: the coarse net's gradient equals that of the coarse-only step (the fine
        depths are detached, nerf/run.py:398), the fine net's is checked on the depths the step used; then 20 Adam steps."""
    import swnerf.embedder as embedder, swnerf.render as render, swnerf.model as model
    from flipcheck import noview_flip_aware_check
    embed_fn, _ = embedder.get_embedder(10, 3, 0)
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=None, netchunk=1024 * 64)
    sds = cases.g12_weights()

    def mk(i, oc=5):
        sd = dict(sds[i])
        sd["output_linear.weight"], sd["output_linear.bias"] = sd["output_linear.weight"][:oc], sd["output_linear.bias"][:oc]
        m = model.vallina_NeRF(**dict(cases.G12_NET, output_ch=oc))
        m.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
        return m.to(dev).train(), sd

    used = lambda net: {k: p.grad for k, p in net.named_parameters() if k.startswith(("pts_linears", "output_linear"))}
    rng = np.random.default_rng(21)
    with torch.enable_grad():
        for n, S, oc, white, chunk_rows, noise_std in ((64, 64, 5, True, 393216, 0.), (37, 40, 5, False, 1024, 1.), (5, 33, 4, True, 64, 0.),
                                                       (3, 2, 5, False, 393216, 0.), (2, 256, 4, True, 393216, 0.5)):
            monkeypatch.setattr(render, "TRAIN_BWD_CHUNK_ROWS", chunk_rows)
            g = cases.g7_inputs(n=n, seed=300 + n)
            rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6.)[:, :8].contiguous()
            tgt = T(rng.uniform(0, 1, (n, 3)).astype(np.float32))
            wa, wd = T(rng.standard_normal(n).astype(np.float32)), T(rng.standard_normal(n).astype(np.float32))
            Graw = T((1e-3 * rng.standard_normal((n, S, oc))).astype(np.float32))

            def ray_loss(ret, idx):                        # a sum over rays: the reference's img2mse + terms on disp, acc and raw
                c = lambda t: t[idx.to(t.device)].to(ret["raw"])
                ok = ~torch.isnan(ret["disp_map"])
                return (((ret["rgb_map"] - c(tgt)) ** 2).sum() / (3 * n) + 0.1 * (ret["acc_map"] * c(wa)).sum() / n
                        + 0.01 * (torch.where(ok, ret["disp_map"], torch.zeros_like(ret["disp_map"])) * c(wd)).sum() / n + (ret["raw"] * c(Graw)).sum())

            kw = dict(retraw=True, N_importance=0, white_bkgd=white, raw_noise_std=noise_std, pytest=True)
            np.random.seed(0)
            noise = T((np.random.rand(n, S) * noise_std).astype(np.float32)) if noise_std > 0 else None     # what pytest=True draws (ray.py:176-184)
            net, sd_np = mk(0, oc)
            assert render.fused_plan(q, [net, None], allow_train=True) == (10, 0, 0)
            hits = []
            monkeypatch.setattr(render, "PASS_HOOK", lambda *a: hits.append(a))
            ret = render.render_rays(rb.to(dev), net, q, S, **kw)
            monkeypatch.setattr(render, "PASS_HOOK", None)
            assert hits and ret["raw"].shape == (n, S, oc) and ret["raw"].requires_grad      # the fused training pass ran
            ray_loss(ret, torch.arange(n)).backward()
            if noise is None:
                o = O.render_rays_generic(rb, lambda e: O.generic_mlp(O.to_torch_sd(sd_np), e, 8, [4], 63, 0, False), S, 0, white_bkgd=white, retraw=True)
                for k in ("rgb_map", "acc_map"):
                    close(ret[k], o[k], atol=2e-5, what=f"training forward {k} (n={n} S={S})")
                close(ret["raw"], o["raw"], atol=1e-3, rtol=1e-4, what="training forward raw")
            z = O.coarse_z(rb[:, 6:7], rb[:, 7:8], S)
            what = f"fused NOVIEW training vs float64 autograd (n={n} S={S} out_ch={oc})"
            fl = noview_flip_aware_check(sd_np, rb, z, white, ray_loss, used(net), what, noise=noise)
            # (b) the generic layer-by-layer path on the same call
            monkeypatch.setenv("SWNERF_TRAIN_OP_PATH", "1")
            net_g, _ = mk(0, oc)
            ret_g = render.render_rays(rb.to(dev), net_g, q, S, **kw)
            monkeypatch.delenv("SWNERF_TRAIN_OP_PATH")
            ray_loss(ret_g, torch.arange(n)).backward()
            close(ret["rgb_map"], ret_g["rgb_map"], atol=2e-5, what="fused vs generic training forward")
            fg = noview_flip_aware_check(sd_np, rb, z, white, ray_loss, used(net_g), what.replace("fused", "generic"), noise=noise)
            print(f"\n[parity] NOVIEW training n={n} S={S} out_ch={oc}: all gradients within 2e-5 of float64; ReLU flips (of risky units) fused {fl}, generic {fg}")
        # (c) the hierarchical step, two nets
        monkeypatch.setattr(render, "TRAIN_BWD_CHUNK_ROWS", 4096)
        g = cases.g12_inputs()
        rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), g["near"], g["far"])[:48, :8].contiguous()
        tgt = T(rng.uniform(0, 1, (48, 3)).astype(np.float32))
        (nc, _), (nf, sd_f) = mk(0), mk(1)
        ret = render.render_rays(rb.to(dev), nc, q, 64, retraw=True, N_importance=128, network_fine=nf, white_bkgd=True)
        assert list(ret.keys()) == ["rgb_map", "disp_map", "acc_map", "raw", "rgb0", "disp0", "acc0", "z_std"] and ret["raw"].shape == (48, 192, 5)
        (torch.mean((ret["rgb_map"] - tgt.to(dev)) ** 2) + torch.mean((ret["rgb0"] - tgt.to(dev)) ** 2)).backward()
        nc2, _ = mk(0)
        r0 = render.render_rays(rb.to(dev), nc2, q, 64, N_importance=0, white_bkgd=True)
        torch.mean((r0["rgb_map"] - tgt.to(dev)) ** 2).backward()
        close(ret["rgb0"], r0["rgb_map"], atol=0, what="rgb0 of the hierarchical step = the coarse-only pass")
        g2 = dict(nc2.named_parameters())
        for k, p in nc.named_parameters():
            if p.grad is not None:
                scale = max(float(g2[k].grad.abs().max()), 1e-12)
                assert float((p.grad - g2[k].grad).abs().max()) <= 2e-6 * scale, f"coarse net {k}"
        with torch.no_grad():
            z_fine = render.render_pass(rb.to(dev), nc2, 64, white_bkgd=True, want=[], n_importance=128)["z_fine"].cpu()
        fine_loss = lambda r, idx: ((r["rgb_map"] - tgt[idx].to(r["raw"])) ** 2).sum() / (3 * 48)
        fl = noview_flip_aware_check(sd_f, rb, z_fine, True, fine_loss, used(nf), "fine net of the hierarchical NOVIEW step on the step's depths")
        print(f"\n[parity] NOVIEW hierarchical step, fine net (48 x 192 rows): within 2e-5 of float64; ReLU flips (of risky units) {fl}")
        # optimizer.step() changes the weights in place: the packed streams (forward and transposed) must follow
        opt = torch.optim.Adam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4)
        before = float(torch.mean((ret["rgb_map"].detach() - tgt.to(dev)) ** 2))
        for _ in range(20):
            opt.zero_grad()
            r = render.render_rays(rb.to(dev), nc, q, 64, N_importance=128, network_fine=nf, white_bkgd=True)
            loss = torch.mean((r["rgb_map"] - tgt.to(dev)) ** 2) + torch.mean((r["rgb0"] - tgt.to(dev)) ** 2)
            loss.backward()
            opt.step()
        after = float(torch.mean((r["rgb_map"].detach() - tgt.to(dev)) ** 2))
        print(f"\n[train] NOVIEW fused, 20 Adam steps on 48 rays: mse {before:.4f} -> {after:.4f}")
        assert after < before
