"""The opt-in bf16x3 render pass (csrc/mlp_core_x3.h, swnerf_render_pass_x3): every weight and activation split into two
bf16 halves, three bf16 MFMAs per product with fp32 accumulation.  NOT the parity path - that is the fp32 pass, checked
against the oracle and the reference's golden vectors in test_gpu_parity.py; here the bf16x3 pass is held against the
fp32 pass of the SAME library on the same inputs, with the tolerances measured on MI355X written next to each gate
(a reduced-precision path must state its PSNR; >= 45 dB asked, > 100 dB measured on equal depths)."""
import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def sw():
    import swnerf.ray, swnerf.embedder, swnerf.model, swnerf.render  # noqa
    import swnerf
    return swnerf


@pytest.fixture(scope="module")
def nets(sw, dev):
    out = []
    for sd in cases.weights_static():
        m = sw.model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict({k: T(v) for k, v in sd.items()})
        out.append(m.to(dev).eval())
    return out


def _rb(sw, dev, n, seed=1):
    g = cases.g7_inputs(n=n, seed=seed)
    return sw.render.pack_ray_batch(T(g["rays_o"]).to(dev), T(g["rays_d"]).to(dev), g["near"], g["far"])


def psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a.double() - b.double()) ** 2)))


WANT = ("rgb_map", "disp_map", "acc_map", "depth_map", "weights", "raw", "z_out")


@pytest.mark.parametrize("n,S,kw", [
    (1024, 64, dict()),                               # whole workgroups, two tiles
    (1021, 64, dict(lindisp=True)),                   # a workgroup with a wave past the last ray (it follows along, stores nothing)
    (3, 40, dict()),                                  # fewer rays than waves, a ragged last tile
    (257, 192, dict(white_bkgd=True)),                # six tiles
    (130, 33, dict(white_bkgd=True, jitter=True, noise=True)),
])
def test_x3_pass_tracks_fp32_pass(sw, dev, nets, n, S, kw):
    """Same sampling, encoding, compositing code; only the MLP arithmetic differs.  Measured on MI355X (C2 scene, 4096 rays):
    max |d raw| 1.2e-4 on values up to 4.7, max |d rgb| 1.1e-5, PSNR 113.8 dB."""
    kw = dict(kw)
    rb = _rb(sw, dev, n)
    gen = torch.Generator(device="cpu").manual_seed(5)
    extra = {}
    if kw.pop("jitter", False):
        extra["t_rand"] = torch.rand((n, S), generator=gen).to(dev)
    if kw.pop("noise", False):
        extra["noise"] = (torch.randn((n, S), generator=gen) * 0.5).to(dev)
    with torch.no_grad():
        a = sw.render.render_pass(rb, nets[1], S, want=WANT, precision="fp32", **kw, **extra)
        b = sw.render.render_pass(rb, nets[1], S, want=WANT, precision="bf16x3", **kw, **extra)
    assert torch.equal(a["z_out"], b["z_out"])                       # the depths never touch the MLP
    scale = float(a["raw"].abs().max())
    assert float((a["raw"] - b["raw"]).abs().max()) < 1e-4 * max(scale, 1.0) + 2e-4, (scale, float((a["raw"] - b["raw"]).abs().max()))
    for k in ("rgb_map", "acc_map", "weights"):
        assert float((a[k] - b[k]).abs().max()) < 1e-4, (k, float((a[k] - b[k]).abs().max()))
    assert float((a["depth_map"] - b["depth_map"]).abs().max()) < 5e-4
    nan = torch.isnan(a["disp_map"])
    assert torch.equal(nan, torch.isnan(b["disp_map"]))
    assert torch.allclose(a["disp_map"][~nan], b["disp_map"][~nan], rtol=1e-3, atol=1e-4)
    if n >= 100:
        assert psnr(a["rgb_map"], b["rgb_map"]) > 95.0


def test_x3_resampling_and_full_render(sw, dev, nets):
    """Coarse pass + hierarchical resampling + fine pass, through render() with the module-level switch.  On equal depths the
    image agrees to > 95 dB; end to end the resampling amplifies last-bit differences of the coarse weights where
    sample_pdf's `denom < 1e-5` branch flips (the same conditioning the fp32 pass shows against the oracle), measured
    61.6 dB on the C2 scene - the gate asks for the reduced-precision bar of SURVEY.md 8d (45 dB) with margin."""
    rb = _rb(sw, dev, 1500, seed=2)
    with torch.no_grad():
        c32 = sw.render.render_pass(rb, nets[0], 64, want=("rgb_map", "weights"), n_importance=128, white_bkgd=True, precision="fp32")
        cx3 = sw.render.render_pass(rb, nets[0], 64, want=("rgb_map", "weights"), n_importance=128, white_bkgd=True, precision="bf16x3")
        assert float((c32["weights"] - cx3["weights"]).abs().max()) < 1e-4
        zf = cx3["z_fine"]
        assert bool((zf[:, 1:] >= zf[:, :-1]).all())                  # the merge still yields sorted depths
        frac = float(((c32["z_fine"] - zf).abs() < 1e-4).float().mean())
        assert frac > 0.98, frac                                        # measured 0.999
        f32 = sw.render.render_pass(rb, nets[1], 192, z_vals=c32["z_fine"], white_bkgd=True, precision="fp32")
        fx3 = sw.render.render_pass(rb, nets[1], 192, z_vals=c32["z_fine"], white_bkgd=True, precision="bf16x3")
        assert psnr(f32["rgb_map"], fx3["rgb_map"]) > 95.0

        K, c2w = cases.synth.lego_camera(40, 56, theta=20.0)
        embed_fn, _ = sw.embedder.get_embedder(10, 3, 0)
        embeddirs_fn, _ = sw.embedder.get_embedder(4, 3, 0)
        query = lambda inputs, viewdirs, network_fn: sw.render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,  # noqa: E731
                                                                           embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
        kw = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets[0], network_query_fn=query, N_samples=64,
                  N_importance=128, network_fine=nets[1], white_bkgd=True, perturb=0., raw_noise_std=0.)
        ref = sw.render.render(40, 56, K, chunk=1024 * 32, c2w=T(c2w).to(dev), **kw)
        prev = sw.render.set_precision("bf16x3")
        try:
            got = sw.render.render(40, 56, K, chunk=1024 * 32, c2w=T(c2w).to(dev), **kw)
        finally:
            sw.render.set_precision(prev)
        assert prev == "fp32" and sw.render.PRECISION == "fp32"
        # "bf16x3-fine": the pass that feeds the resampling stays fp32 -> rgb0 / z_std bit-identical, the image > 95 dB
        sw.render.set_precision("bf16x3-fine")
        try:
            mix = sw.render.render(40, 56, K, chunk=1024 * 32, c2w=T(c2w).to(dev), **kw)
        finally:
            sw.render.set_precision("fp32")
        assert torch.equal(mix[3]["rgb0"], ref[3]["rgb0"]) and torch.equal(mix[3]["z_std"], ref[3]["z_std"])
        assert not torch.equal(mix[0], ref[0]) and psnr(ref[0], mix[0]) > 95.0
        assert not torch.equal(ref[0], got[0])                          # it really ran the other path
        assert psnr(ref[0], got[0]) > 50.0
        assert psnr(ref[3]["rgb0"], got[3]["rgb0"]) > 95.0              # the coarse image has no resampling in front of it


def test_plain_bf16_is_a_different_class(sw, dev, nets):
    """terms = 1 (operands rounded to bf16, 8 significant bits): kept as the yardstick that shows what the split buys -
    36.7 dB on the C2 scene against 113.8 dB for bf16x3 - and as a smoke test of the shared kernel body."""
    rb = _rb(sw, dev, 2048)
    with torch.no_grad():
        a = sw.render.render_pass(rb, nets[1], 64, white_bkgd=True, precision="fp32")
        b = sw.render.render_pass(rb, nets[1], 64, white_bkgd=True, precision="bf16")
        c = sw.render.render_pass(rb, nets[1], 64, white_bkgd=True, precision="bf16x3")
    p1, p3 = psnr(a["rgb_map"], b["rgb_map"]), psnr(a["rgb_map"], c["rgb_map"])
    assert 25.0 < p1 < 60.0 and p3 > p1 + 40.0, (p1, p3)


def test_x3_blob_follows_the_weights(sw, dev, nets):
    """packed_x3() is cached on the parameters' versions like packed(): an optimizer step must invalidate it."""
    net = sw.model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    net.load_state_dict(nets[1].state_dict())
    net = net.to(dev).eval()
    rb = _rb(sw, dev, 64)
    with torch.no_grad():
        a = sw.render.render_pass(rb, net, 64, precision="bf16x3")["rgb_map"].clone()
        blob = net.packed_x3()[0]
        assert net.packed_x3()[0] is blob
        net.rgb_linear.bias.add_(0.25)
        b = sw.render.render_pass(rb, net, 64, precision="bf16x3")["rgb_map"]
        ref = sw.render.render_pass(rb, net, 64, precision="fp32")["rgb_map"]
    assert net.packed_x3()[0] is not blob
    assert float((a - b).abs().max()) > 1e-3 and float((b - ref).abs().max()) < 1e-4


def test_x3_dnerf_tracks_fp32_pass(sw, dev):
    """DirectTemporalNeRF (deformation net, re-embedding of x + dx, canonical net) in one bf16x3 pass, t = 0.5, and the
    `t == 0 and zero_canonical` branch (canonical net alone).  Measured on MI355X: t = 0.5: max |d rgb| 1.2e-4, 96 dB;
    t = 0: 8.8e-6, 115 dB."""
    import swnerf.render_dnerf  # noqa
    e10, _ = sw.embedder.get_embedder(10, 3, 0)
    dn = sw.model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27,
                                   input_ch_time=21, use_viewdirs=True, embed_fn=e10, zero_canonical=True)
    dn.load_state_dict({k: T(v) for k, v in cases.weights_dnerf().items()})
    dn = dn.to(dev).eval()
    g = cases.g8_inputs(n=515)
    want = ("rgb_map", "acc_map", "raw", "dx", "weights")
    for t, deform in ((0.5, True), (0.0, False)):
        rb = sw.render.pack_ray_batch(T(g["rays_o"]).to(dev), T(g["rays_d"]).to(dev), g["near"], g["far"], frame_time=t)
        with torch.no_grad():
            a = sw.render.render_pass(rb, dn, 64, want=want, white_bkgd=True, run_deform=deform, n_importance=128, precision="fp32")
            b = sw.render.render_pass(rb, dn, 64, want=want, white_bkgd=True, run_deform=deform, n_importance=128, precision="bf16x3")
            fa = sw.render.render_pass(rb, dn, 192, z_vals=a["z_fine"], want=want, white_bkgd=True, run_deform=deform, precision="fp32")
            fb = sw.render.render_pass(rb, dn, 192, z_vals=a["z_fine"], want=want, white_bkgd=True, run_deform=deform, precision="bf16x3")
        for x, y in ((a, b), (fa, fb)):
            assert float((x["dx"] - y["dx"]).abs().max()) < 5e-5, float((x["dx"] - y["dx"]).abs().max())
            if not deform:
                assert float(y["dx"].abs().max()) == 0.0
            # the canonical net sees gamma(x + dx) up to the band 2^9: a 4e-6 difference in dx is a 2e-3 phase difference
            # there, so raw moves by ~1e-3 where the static net's moves by 1e-4 (the fp32 pass shows the same
            # conditioning against the CPU oracle: 62 dB on C5 against 81 dB on C4)
            scale = max(float(x["raw"].abs().max()), 1.0)
            m = dict(raw=float((x["raw"] - y["raw"]).abs().max()), **{k: float((x[k] - y[k]).abs().max()) for k in ("rgb_map", "acc_map", "weights")},
                     psnr=psnr(x["rgb_map"], y["rgb_map"]))
            print(f"t={t} S={x['raw'].shape[1]}: {m}")
            assert m["raw"] < (1e-3 if deform else 1e-4) * scale, m
            assert max(m["rgb_map"], m["acc_map"], m["weights"]) < (5e-4 if deform else 1e-4), m
            assert m["psnr"] > (80.0 if deform else 95.0), m
        assert bool((b["z_fine"][:, 1:] >= b["z_fine"][:, :-1]).all())


def test_x3_argument_checks(sw, dev):
    from swnerf import _lib
    a = _lib.PassArgs()
    buf = torch.zeros(64, device=dev)
    a.ray_batch, a.n_rays, a.cols, a.kind, a.packed, a.n_samples = buf.data_ptr(), 1, 11, _lib.NET_DNERF, buf.data_ptr(), 64
    assert _lib.lib().swnerf_render_pass_x3(a, 3, None) == -1 and b"frame_time" in _lib.lib().swnerf_last_error()
    a.kind = 7
    assert _lib.lib().swnerf_render_pass_x3(a, 3, None) == -1
    a.kind = _lib.NET_CANON
    assert _lib.lib().swnerf_render_pass_x3(a, 2, None) == -1 and b"terms" in _lib.lib().swnerf_last_error()
    assert _lib.lib().swnerf_packed_x3_floats_kind(7) == 0


# ---- against the REFERENCE: the golden fixtures (round-2 VERDICT: the tests above compare the bf16x3 pass with this library's
# own fp32 pass only).  Floors = measured on MI355X (printed below) - 3 dB, never under the 45 dB SURVEY.md 8d asks of a
# reduced-precision path.  The fp32 pass's own PSNR against the same goldens (test_gpu_parity.py) is the yardstick: C2
# 78.9 dB, D-NeRF t = 0.5 57.7 dB.
# measured (round 3): C2 fp32 78.9 / bf16x3 59.9 / bf16x3-fine 78.9 dB; D-NeRF t = 0.5 (512 rays) fp32 56.1 / bf16x3 50.2 / bf16x3-fine 56.1 dB
X3_GOLDEN_FLOORS = {("C2", "bf16x3"): 56.9, ("C2", "bf16x3-fine"): 75.9, ("dnerf t=0.5", "bf16x3"): 47.2, ("dnerf t=0.5", "bf16x3-fine"): 53.1}


def test_x3_against_the_reference_goldens(sw, dev, nets, golden):
    import swnerf.render_dnerf  # noqa
    embed_fn, _ = sw.embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = sw.embedder.get_embedder(4, 3, 0)
    embedtime_fn, _ = sw.embedder.get_embedder(10, 1, 0)
    q = lambda inputs, viewdirs, network_fn: sw.render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
    qd = lambda inputs, viewdirs, ts, network_fn: sw.render_dnerf.run_network(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn,
                                                                             embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn, embd_time_discr=True)
    dn = sw.model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27,
                                   input_ch_time=21, use_viewdirs=True, embed_fn=embed_fn, zero_canonical=True)
    dn.load_state_dict({k: T(v) for k, v in cases.weights_dnerf().items()})
    dn = dn.to(dev).eval()
    from oracle import nerf_oracle as O
    g7, g8 = cases.g7_inputs(), cases.g8_inputs()
    rb7 = O.make_ray_batch(T(g7["rays_o"]), T(g7["rays_d"]), g7["near"], g7["far"]).to(dev)
    rb8 = O.make_ray_batch(T(g8["rays_o"]), T(g8["rays_d"]), g8["near"], g8["far"], frame_time=0.5).to(dev)
    ref7, ref8 = golden("g7_c2"), golden("g8_dnerf_t5")
    measured = {}
    for mode in ("fp32", "bf16x3", "bf16x3-fine"):
        prev = sw.render.set_precision(mode)
        try:
            with torch.no_grad():
                a = sw.render.render_rays(rb7, nets[0], q, 64, N_importance=128, network_fine=nets[1], white_bkgd=True)
                b = sw.render_dnerf.render_rays(rb8, dn, qd, 64, N_importance=128, white_bkgd=True)
        finally:
            sw.render.set_precision(prev)
        measured[("C2", mode)] = psnr(a["rgb_map"].cpu(), T(ref7["rgb_map"]))
        measured[("dnerf t=0.5", mode)] = psnr(b["rgb_map"].cpu(), T(ref8["rgb_map"]))
        if mode == "bf16x3-fine":                      # the coarse pass stayed fp32: its outputs are the parity path's, to the reference's 2e-5
            assert float((a["rgb0"].cpu() - T(ref7["rgb0"])).abs().max()) <= 2e-5
    for k, v in measured.items():
        print(f"\n[x3 vs golden] {k[0]}, {k[1]}: PSNR of rgb_map against the REFERENCE's render {v:.1f} dB")
    for k, floor in X3_GOLDEN_FLOORS.items():
        assert measured[k] >= max(floor, 45.0), (k, measured[k], floor)
