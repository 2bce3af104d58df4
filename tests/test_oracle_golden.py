"""Pins oracle/nerf_oracle.py to the reference: every function of the CPU restatement
is run on the seeded inputs of tests/golden/cases.py and compared with the outputs the
unmodified reference produced for them (tests/golden/*.npz, made by make_golden.py).

Same torch CPU kernels, same op order => the comparison is (near) bit-exact; the
tolerances below only absorb the thread-count dependence of MKL sgemm blocking."""
import numpy as np
import torch
import pytest

import cases
from oracle import nerf_oracle as O

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


def close(a, b, atol=0.0, rtol=0.0):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, equal_nan=True)


def test_g1_embed(golden):
    g, ref = cases.g1_inputs(), golden("g1_embed")
    assert ref["crc"] == cases.checksum(g["pts"], g["dirs"], g["t"])
    close(O.embed(T(g["pts"]), 10), ref["pts"])
    close(O.embed(T(g["dirs"]), 4), ref["dirs"])
    close(O.embed(T(g["t"]), 10), ref["t"])
    assert O.embed_dim(10, 3) == 63 and O.embed_dim(4, 3) == 27 and O.embed_dim(10, 1) == 21


def test_g2_rays(golden):
    g, ref = cases.g2_inputs(), golden("g2_rays")
    st = int(ref["step"][0])
    o, d = O.get_rays(400, 400, g["K400"], g["c2w400"])
    close(d.reshape(-1, 3)[::st], ref["d_k"])
    close(o.reshape(-1, 3)[::st], ref["o_k"])
    assert o.stride()[:2] == (0, 0)            # stride-0 expand like ray.py:37
    _, d = O.get_rays(400, 400, g["focal400"], g["c2w400"])
    close(d.reshape(-1, 3)[::st], ref["d_f"])
    o, d = O.get_rays(32, 48, g["K_small"], g["c2w_small"])
    close(d.reshape(-1, 3), ref["d_s"])
    close(o.reshape(-1, 3), ref["o_s"])
    on, dn = O.get_rays_np(400, 400, g["K400"], g["c2w400"])
    assert str(dn.dtype) == str(ref["np_dtype"][0])
    close(dn.reshape(-1, 3)[::st], ref["d_np"])
    o, d = O.get_rays(378, 504, g["Kf"], g["c2wf"])
    close(d.reshape(-1, 3)[::st], ref["d_l"])
    o2, d2 = O.ndc_rays(378, 504, g["Kf"][0][0], 1., o, d)
    close(o2.reshape(-1, 3)[::st], ref["o_ndc"])
    close(d2.reshape(-1, 3)[::st], ref["d_ndc"])


def test_g3_coarse(golden):
    g, ref = cases.g3_inputs(), golden("g3_coarse")
    o, d = T(g["rays_o"]), T(g["rays_d"])
    for lindisp in (False, True):
        for perturb in (0, 1):
            z = O.coarse_z(T(g["near"]), T(g["far"]), 64, lindisp, T(g["t_rand"]) if perturb else None)
            pts = o[..., None, :] + d[..., None, :] * z[..., :, None]
            close(pts[:32], ref[f"pts_l{int(lindisp)}_p{perturb}"])


def test_g4_mlp(golden):
    g, ref = cases.g4_inputs(), golden("g4_mlp")
    sd_c, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    x = T(g["x"])
    close(O.nerf_mlp(sd_c, x), ref["vanilla"], atol=2e-6, rtol=1e-5)
    close(O.nerf_mlp(sd_f, x), ref["original"], atol=2e-6, rtol=1e-5)
    sd_d = O.to_torch_sd(cases.weights_dnerf())
    for tv in (0.0, 0.5):
        te = O.embed(torch.full((x.shape[0], 1), tv), 10)
        out, dx = O.dnerf_mlp(sd_d, x, te)
        close(out, ref[f"dn_out_t{int(tv*10)}"], atol=2e-6, rtol=1e-5)
        close(dx, ref[f"dn_dx_t{int(tv*10)}"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("S", [64, 192])
def test_g5_raw2outputs(golden, S):
    g, ref = cases.g5_inputs(S), golden(f"g5_raw2outputs_S{S}")
    for wb in (False, True):
        r = O.raw2outputs(T(g["raw"]), T(g["z"]), T(g["rays_d"]), 0., wb)
        for k, v in zip(["rgb", "disp", "acc", "weights", "depth"], r):
            close(v, ref[f"{k}_w{int(wb)}"])
    assert np.isnan(ref["disp_w0"][0]) and ref["acc_w0"][0] == 0     # the empty ray
    r = O.raw2outputs(T(g["raw"]), T(g["z"]), T(g["rays_d"]), 1.0, True, noise=T(g["noise"]) * 1.0)
    for k, v in zip(["rgb", "disp", "acc", "weights", "depth"], r):
        close(v, ref[f"{k}_noise"])


def test_g6_sample_pdf(golden):
    g, ref = cases.g6_inputs(), golden("g6_sample_pdf")
    s_det = O.sample_pdf(T(g["bins"]), T(g["weights"]), 128, det=True)
    s_rnd = O.sample_pdf(T(g["bins"]), T(g["weights"]), 128, u=T(g["u"]))
    close(s_det, ref["det"])
    close(s_rnd, ref["rnd"])
    close(torch.sort(torch.cat([T(g["z"]), s_det], -1), -1)[0], ref["z_det"])
    close(torch.std(s_rnd, dim=-1, unbiased=False), ref["std_rnd"])


def _rb(g, t=None):
    return O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), g["near"], g["far"], frame_time=t)


def _cmp_dict(ret, ref, keys, atol, rtol=1e-4, nraw=32):
    for k in keys:
        v = ret[k]
        if k in ("raw", "position_delta"):
            v = v[:nraw]
        close(v, ref[k], atol=atol, rtol=rtol)


def test_g7_static_render_rays(golden):
    sd_c, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    g = cases.g7_inputs()
    ref = golden("g7_c1")
    r = O.render_rays(_rb(g), sd_c, None, 64, 0, white_bkgd=True, retraw=True)
    _cmp_dict(r, ref, ["rgb_map", "disp_map", "acc_map", "raw"], atol=2e-6)
    ref = golden("g7_c2")
    r = O.render_rays(_rb(g), sd_c, sd_f, 64, 128, white_bkgd=True, retraw=True)
    _cmp_dict(r, ref, ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std", "raw"], atol=5e-6)
    gs = cases.g7_inputs(n=256, seed=11)
    ref = golden("g7_lindisp")
    r = O.render_rays(_rb(gs), sd_c, None, 64, 128, lindisp=True, white_bkgd=False)
    _cmp_dict(r, ref, ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"], atol=5e-6)
    ref = golden("g7_perturb")
    rnd = cases.g7_rand_inputs(256)
    r = O.render_rays(_rb(gs), sd_c, sd_f, 64, 128, white_bkgd=True, t_rand=T(rnd["t_rand"]), u=T(rnd["u"]))
    _cmp_dict(r, ref, ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"], atol=5e-6)


def test_g7_ndc_render(golden):
    sd_c, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    g, ref = cases.g7_ndc_inputs(), golden("g7_ndc")
    rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 0., 1., ndc=True, H=g["H"], W=g["W"], focal=g["focal"])
    r = O.render_rays(rb, sd_c, sd_f, 64, 128, white_bkgd=False)
    _cmp_dict(r, ref, ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"], atol=5e-6)


def test_g8_dnerf_render_rays(golden):
    sd = O.to_torch_sd(cases.weights_dnerf())
    g = cases.g8_inputs()
    for tv in (0.0, 0.5):
        ref = golden(f"g8_dnerf_t{int(tv*10)}")
        r = O.render_rays_dnerf(_rb(g, tv), sd, 64, 128, white_bkgd=True, retraw=True)
        _cmp_dict(r, ref, ["rgb_map", "disp_map", "acc_map", "z_vals", "z_std", "raw", "position_delta"], atol=5e-6)
        assert set(ref.keys()) - {"crc"} == {"rgb_map", "disp_map", "acc_map", "z_vals", "position_delta", "raw", "z_std"}
    ref = golden("g8_dnerf_coarse_only")
    r = O.render_rays_dnerf(_rb(g, 0.25)[:128], sd, 64, 0, white_bkgd=True)
    _cmp_dict(r, ref, ["rgb_map", "disp_map", "acc_map", "z_vals", "position_delta"], atol=5e-6)


def test_g10_mesh_grid_query(golden):
    """nerf/extract_mesh.py sample_grid + generate_viewdirs + the 2-D network_query_fn."""
    from swnerf import mesh
    ref = golden("g10_mesh_query")
    vd = mesh.generate_viewdirs(100)
    close(vd, ref["viewdirs100"])
    _, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    R = cases.G10_RES
    ax = [np.linspace(b[0], b[1], R) for b in cases.G10_BOUNDS]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    assert np.array_equal(X, ref["X"])
    pts = torch.tensor(np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1), dtype=torch.float32)
    dirs = torch.tensor(mesh.generate_viewdirs(cases.G10_VIEWS), dtype=torch.float32)
    dens, col = O.sample_grid_points(sd_f, pts, dirs)
    close(dens.reshape(R, R, R), ref["density"], atol=2e-6, rtol=1e-5)
    close(col.reshape(R, R, R, 3), ref["color"], atol=2e-6, rtol=1e-5)
    one = O.query_points(sd_f, pts[:40], torch.tensor(np.tile(vd[3][None], (40, 1)), dtype=torch.float32))
    close(one, ref["per_point_raw"], atol=2e-6, rtol=1e-5)


def test_g11_generic_shapes(golden):
    """The oracle's any-shape restatements (use_viewdirs=False, other D / W / skips, D-NeRF at D=4) against the
    reference's outputs (tests/golden/make_golden_generic.py)."""
    ref = golden("g11_generic")
    g = cases.g11_inputs()
    assert int(ref["checksum"][0]) == int(cases.checksum(g["pts"], g["dirs"], g["rays"]["rays_o"], g["rays"]["rays_d"])[0])
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for name, kw in cases.G11_NETS.items():
        sd = O.to_torch_sd(cases.g11_weights(name))
        L = (kw["input_ch"] // 3 - 1) // 2
        x = O.embed(T(g["pts"]), L)
        if kw["input_ch_views"]:
            x = torch.cat([x, O.embed(T(g["dirs"]), 4)], -1)
        out = O.generic_mlp(sd, x, kw["D"], kw["skips"], kw["input_ch"], kw["input_ch_views"], kw["use_viewdirs"])
        np.testing.assert_allclose(out.numpy(), ref[f"mlp_{name}"], atol=2e-5, rtol=1e-5)
    kw = cases.G11_DNERF
    sd = O.to_torch_sd(cases.g11_dnerf_weights())
    x = torch.cat([O.embed(T(g["pts"]), 10), O.embed(T(g["dirs"]), 4)], -1)
    for tv in (0.0, 0.5):
        te = O.embed(torch.full((x.shape[0], 1), tv), 10)
        out, dx = O.generic_dnerf_mlp(sd, x, te, kw["D"], kw["skips"], kw["input_ch"], kw["input_ch_views"], kw["use_viewdirs"])
        np.testing.assert_allclose(dx.numpy(), ref[f"dnerf_dx_t{int(tv*10)}"], atol=1e-6)
        np.testing.assert_allclose(out.numpy(), ref[f"dnerf_out_t{int(tv*10)}"], atol=2e-4, rtol=1e-4)
    kwn = cases.G11_NETS["novd"]
    sdn = O.to_torch_sd(cases.g11_weights("novd"))
    r = g["rays"]
    o_, d_ = T(r["rays_o"]), T(r["rays_d"])
    rb = torch.cat([o_, d_, 2. * torch.ones_like(d_[:, :1]), 6. * torch.ones_like(d_[:, :1])], -1)
    ret = O.render_rays_generic(rb, lambda e: O.generic_mlp(sdn, e, kwn["D"], kwn["skips"], kwn["input_ch"], 0, False), 32, 32, white_bkgd=True)
    for k in ("rgb0", "acc0", "rgb_map", "acc_map", "z_std"):
        np.testing.assert_allclose(ret[k].numpy(), ref[f"rr_{k}"], atol=2e-5 if k.endswith("0") else 5e-4, err_msg=k)


def test_g12_render_rays_without_viewdirs(golden):
    """The oracle's render_rays without view directions (two nets, output_ch 5) against the reference's own
    nerf/run.py render_rays on NON-degenerate weights (tests/golden/make_golden_noview.py): the coarse-only outputs and
    raw to 1e-5-level, the image behind the hierarchical resampling to the conditioning of sample_pdf (the oracle runs the
    same ATen CPU kernels as the reference, so it is bit-close here too)."""
    ref = golden("g12_noview")
    g = cases.g12_inputs()
    ws = cases.g12_weights()
    assert int(ref["checksum"][0]) == int(cases.checksum(g["rays_o"], g["rays_d"], *[v for sd in ws for v in sd.values()])[0])
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    o_, d_ = T(g["rays_o"]), T(g["rays_d"])
    rb = torch.cat([o_, d_, g["near"] * torch.ones_like(d_[:, :1]), g["far"] * torch.ones_like(d_[:, :1])], -1)
    fns = [(lambda e, sd=O.to_torch_sd(sd_np): O.generic_mlp(sd, e, 8, [4], 63, 0, False)) for sd_np in ws]
    with torch.no_grad():
        c = O.render_rays_generic(rb, fns[0], 64, 0, white_bkgd=True, retraw=True)
        h = O.render_rays_two_nets_generic(rb, fns[0], fns[1], 64, 128, white_bkgd=True)
    assert 0.2 < float(ref["c_acc_map"].mean()) < 0.8 and float(ref["c_rgb_map"].std()) > 0.05      # the case is not vacuous
    for k in ("rgb_map", "disp_map", "acc_map"):
        np.testing.assert_allclose(c[k].numpy(), ref[f"c_{k}"], atol=2e-5, rtol=1e-4, err_msg=k)
    np.testing.assert_allclose(c["raw"][:16].numpy(), ref["c_raw"], atol=2e-4, rtol=1e-4)
    assert tuple(ref["h_raw_shape"]) == (256, 192, 5)
    for k in ("rgb0", "disp0", "acc0"):
        np.testing.assert_allclose(h[k].numpy(), ref[f"h_{k}"], atol=2e-5, rtol=1e-4, err_msg=k)
    np.testing.assert_allclose(h["z_std"].numpy(), ref["h_z_std"], atol=1e-4)
    for k in ("rgb_map", "acc_map"):
        d = np.abs(h[k].numpy() - ref[f"h_{k}"])
        assert float((d <= 2e-4).mean()) >= 0.97 and float(d.max()) <= 2e-2, (k, float((d <= 2e-4).mean()), float(d.max()))
