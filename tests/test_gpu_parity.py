"""GPU parity tests: the HIP path (through the C ABI, via the reference-mirroring Python
modules) against (a) the golden vectors captured from the reference and (b) the CPU oracle
on the same seeded inputs.  fp32 everywhere; tolerances are stated per test and were set
from the deltas measured on MI355X (DESIGN.md "Parity").

What differs from the reference's CPU arithmetic, and so bounds the tolerance:
  * GEMM accumulation order (MFMA k-ordered fmaf chain vs MKL sgemm blocking)   ~1e-6 rel/layer
  * sin/cos (own Cody-Waite + Cephes poly, 9.3e-8 abs) vs Sleef                  ~1e-7 abs
  * float reductions over samples (wave tree vs ATen vectorised)                ~1e-7 rel
"""
import numpy as np
import pytest
import torch

import cases
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def sw():
    import swnerf.ray, swnerf.embedder, swnerf.model, swnerf.render, swnerf.render_dnerf  # noqa
    import swnerf
    return swnerf


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol, equal_nan=True, err_msg=what)


def close_mostly(a, b, atol, frac, hard, what="", rel=False):
    """At least `frac` of the elements within atol (relative to max(1,|b|) if rel), every element
    within `hard`.  For quantities downstream of sample_pdf, whose (u - cdf_lo)/denom has the
    reference's own discontinuity `denom < 1e-5 -> 1` (ray.py:148-149): the pdf of an empty bin of
    an opaque ray is 1e-5/(sum w + 62e-5) ~ 0.9994e-5 and the float cdf is quantised at 6e-8 near
    1.0, so `denom` lands on either side of 1e-5 by rounding alone.  Any float reordering (also the
    reference's own CPU-vs-GPU difference) therefore moves a few samples by up to one bin width
    inside bins that carry < 1e-5 of the mass (DESIGN.md "Parity")."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
    d = np.abs(np.nan_to_num(a) - np.nan_to_num(b))
    if rel:
        d = d / np.maximum(1.0, np.abs(np.nan_to_num(b)))
    ok = float((d <= atol).mean()) if d.size else 1.0
    assert ok >= frac, f"{what}: only {ok:.4f} of elements within {atol} (need {frac})"
    assert d.size == 0 or float(d.max()) <= hard, f"{what}: max |delta| {d.max():.3e} > {hard}"


def psnr(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(-10 * np.log10(max(float(((a - b) ** 2).mean()), 1e-20)))


def maxdiff(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    m = ~(np.isnan(a) | np.isnan(b))
    return float(np.abs(a[m] - b[m]).max()) if m.any() else 0.0


# ------------------------------------------------------------------------------- embedder.py
def test_embed_golden(sw, dev, golden):
    g, ref = cases.g1_inputs(), golden("g1_embed")
    e10, d10 = sw.embedder.get_embedder(10, 3, 0)
    e4, d4 = sw.embedder.get_embedder(4, 3, 0)
    et, dt = sw.embedder.get_embedder(10, 1, 0)
    assert (d10, d4, dt) == (63, 27, 21)
    # |arg| up to 6*512: own sin/cos is within 1.2e-7 abs of libm; Sleef within 1 ulp
    close(e10(T(g["pts"]).to(dev)), ref["pts"], atol=3e-7, what="embed pts")
    close(e4(T(g["dirs"]).to(dev)), ref["dirs"], atol=3e-7, what="embed dirs")
    close(et(T(g["t"]).to(dev)), ref["t"], atol=3e-7, what="embed t")
    ident, d = sw.embedder.get_embedder(10, 3, -1)
    assert d == 3 and torch.equal(ident(T(g["pts"])), T(g["pts"]))
    # leading batch dims and an empty input
    x = T(g["pts"]).to(dev).reshape(4, 256, 3)
    assert e10(x).shape == (4, 256, 63)
    assert e10(torch.empty((0, 3), device=dev)).shape == (0, 63)


# ------------------------------------------------------------------------------------ ray.py
def test_get_rays_ndc_golden(sw, dev, golden):
    g, ref = cases.g2_inputs(), golden("g2_rays")
    st = int(ref["step"][0])
    c2w = T(g["c2w400"]).to(dev)
    o, d = sw.ray.get_rays(400, 400, g["K400"], c2w)
    assert o.shape == d.shape == (400, 400, 3) and o.stride()[:2] == (0, 0)
    close(d.reshape(-1, 3)[::st], ref["d_k"], atol=1e-7, rtol=1e-6, what="get_rays K")
    close(o.reshape(-1, 3)[::st], ref["o_k"], atol=0, what="rays_o")
    _, d = sw.ray.get_rays(400, 400, g["focal400"], c2w)
    close(d.reshape(-1, 3)[::st], ref["d_f"], atol=1e-7, rtol=1e-6, what="get_rays focal")
    o, d = sw.ray.get_rays(32, 48, g["K_small"], T(g["c2w_small"]).to(dev))
    close(d.reshape(-1, 3), ref["d_s"], atol=1e-7, rtol=1e-6, what="get_rays small")
    close(o.reshape(-1, 3), ref["o_s"], atol=0)
    on, dn = sw.ray.get_rays_np(400, 400, g["K400"], g["c2w400"])
    close(dn.reshape(-1, 3)[::st], ref["d_np"], atol=0, what="get_rays_np")
    o, d = sw.ray.get_rays(378, 504, g["Kf"], T(g["c2wf"]).to(dev))
    o2, d2 = sw.ray.ndc_rays(378, 504, g["Kf"][0][0], 1., o, d)
    close(o2.reshape(-1, 3)[::st], ref["o_ndc"], atol=2e-7, rtol=2e-6, what="ndc o")
    close(d2.reshape(-1, 3)[::st], ref["d_ndc"], atol=2e-7, rtol=2e-6, what="ndc d")
    # a row range equals the same rows of the full grid (the per-rank entry of the sharded render)
    _, dr = sw.ray.get_rays_range(400, 400, g["K400"], c2w, 400 * 100 + 7, 1000)
    _, dfull = sw.ray.get_rays(400, 400, g["K400"], c2w)
    assert torch.equal(dr, dfull.reshape(-1, 3)[400 * 100 + 7:400 * 100 + 1007])


def test_sample_coarse_golden(sw, dev, golden):
    """a5 as its own op (nerf/run.py:355-385): z_vals and pts against the reference's values (G3): linear and lindisp
    spacing, with and without the stratified jitter (injected t_rand); and == the torch-op form of the same lines."""
    g, ref = cases.g3_inputs(), golden("g3_coarse")
    rb = O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), 2., 6.).to(dev)
    for lindisp in (False, True):
        for perturb in (0, 1):
            tr = T(g["t_rand"]).to(dev) if perturb else None
            z, pts = sw.render.sample_coarse(rb, 64, lindisp, tr, want_pts=True)
            assert z.shape == (256, 64) and pts.shape == (256, 64, 3)
            close(pts[:32], ref[f"pts_l{int(lindisp)}_p{perturb}"], atol=2e-6, what=f"pts lindisp={lindisp} perturb={perturb}")
            zo = O.coarse_z(T(g["near"]), T(g["far"]), 64, lindisp, T(g["t_rand"]) if perturb else None)
            close(z, zo, atol=1e-6, what="z_vals")
    assert sw.render.sample_coarse(rb[:0], 64).shape == (0, 64)
    z7 = sw.render.sample_coarse(rb[:5], 7)                  # odd sample count: the two-sided linspace formula
    close(z7, O.coarse_z(T(g["near"][:5]), T(g["far"][:5]), 7), atol=0.0, what="S=7")


@pytest.mark.parametrize("S", [64, 192])
def test_raw2outputs_golden(sw, dev, golden, S):
    g, ref = cases.g5_inputs(S), golden(f"g5_raw2outputs_S{S}")
    raw, z, d = (T(g[k]).to(dev) for k in ("raw", "z", "rays_d"))
    for wb in (False, True):
        r = sw.ray.raw2outputs(raw, z, d, 0, wb)
        for k, v in zip(["rgb", "disp", "acc", "weights", "depth"], r):
            close(v, ref[f"{k}_w{int(wb)}"], atol=2e-6, rtol=2e-5, what=f"raw2outputs {k} S={S}")
    assert torch.isnan(r[1][0]) and float(r[2][0]) == 0.0          # empty ray: disp NaN, acc 0
    r = sw.ray.raw2outputs(raw, z, d, 1.0, True, pytest=True)
    for k, v in zip(["rgb", "disp", "acc", "weights", "depth"], r):
        close(v, ref[f"{k}_noise"], atol=2e-6, rtol=2e-5, what=f"raw2outputs noise {k}")


def test_raw2outputs_ragged(sw, dev):
    """S not a multiple of the wave width, S=1, N not a multiple of 4, N=0."""
    rng = np.random.default_rng(7)
    for N, S in ((5, 2), (3, 33), (7, 100), (1, 257)):
        raw = T(rng.standard_normal((N, S, 4)).astype(np.float32))
        z = T(np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1))
        d = T(rng.standard_normal((N, 3)).astype(np.float32))
        ref = O.raw2outputs(raw, z, d, 0., True)
        got = sw.ray.raw2outputs(raw.to(dev), z.to(dev), d.to(dev), 0, True)
        for a, b in zip(got, ref):
            close(a, b, atol=2e-6, rtol=2e-5, what=f"ragged N={N} S={S}")
    got = sw.ray.raw2outputs(torch.empty((0, 8, 4), device=dev), torch.empty((0, 8), device=dev), torch.empty((0, 3), device=dev))
    assert got[0].shape == (0, 3) and got[3].shape == (0, 8)
    with pytest.raises(RuntimeError, match="degenerate"):      # the reference yields an EMPTY weights tensor for S=1
        sw.ray.raw2outputs(torch.zeros((2, 1, 4), device=dev), torch.zeros((2, 1), device=dev), torch.ones((2, 3), device=dev))


MAX_FLIPS_G6 = {"det": 16, "rnd": 8}        # of 16 384 samples each; measured on MI355X: 7 and 2


def test_sample_pdf_golden(sw, dev, golden):
    g, ref = cases.g6_inputs(), golden("g6_sample_pdf")
    bins, w = T(g["bins"]).to(dev), T(g["weights"]).to(dev)
    s_det = sw.ray.sample_pdf(bins, w, 128, det=True)
    s_rnd = sw.ray.sample_pdf(bins, w, 128, det=False, pytest=True)
    # hard bound = the width of the bin each reference sample was drawn in: a sample whose bin flips across the denom
    # threshold stays in its bin; and the NUMBER of such samples is bounded (measured on MI355X: see the print)
    for name, got_s in (("det", s_det), ("rnd", s_rnd)):
        bound = bin_width_bound(got_s.cpu().numpy(), ref[name], g["bins"])
        dlt = np.abs(got_s.cpu().numpy().astype(np.float64) - ref[name])
        assert np.all(dlt <= bound * 1.0001 + 1e-6), f"sample_pdf {name}: a sample left the bin it was drawn in"
        nfl = flips(got_s, ref[name])
        print(f"\n[parity] sample_pdf {name}: {nfl} of {dlt.size} samples differ by > 2e-5; max {dlt.max():.3e} (bin widths {bound.min():.3f}..{bound.max():.3f})")
        assert nfl <= MAX_FLIPS_G6[name], f"sample_pdf {name}: {nfl} flipped samples"
        close_mostly(got_s, ref[name], atol=2e-5, frac=0.995, hard=float(bound.max()) + 1e-6, what=f"sample_pdf {name}")
    # the well-conditioned statement of the same thing: cdf(sample) == u.  Evaluate the reference
    # piecewise-linear cdf (float64) at our samples and compare with the u that produced them.
    wn = g["weights"].astype(np.float64) + 1e-5
    cdf = np.concatenate([np.zeros((128, 1)), np.cumsum(wn / wn.sum(-1, keepdims=True), -1)], -1)
    for got, u in ((s_det, np.broadcast_to(np.linspace(0., 1., 128), (128, 128))), (s_rnd, g["u"].astype(np.float64))):
        got = got.cpu().numpy().astype(np.float64)
        for r in range(128):
            back = np.interp(got[r], g["bins"][r].astype(np.float64), cdf[r])
            flat = np.interp(got[r] + 1e-4, g["bins"][r].astype(np.float64), cdf[r]) - back < 1e-5 * 1e-4 / 0.06
            # bins whose mass is < 1e-5 snap to the bin edge by design (ray.py:148-149): skip them
            assert np.all((np.abs(back - u[r]) < 2e-5) | flat), f"cdf(sample) != u on row {r}"
    # the fused variant: samples + sort(cat[z, samples]) + std
    from swnerf import _lib
    z = T(g["z"]).to(dev)
    smp = torch.empty((128, 128), device=dev)
    zs = torch.empty((128, 192), device=dev)
    sd = torch.empty((128,), device=dev)
    _lib.check(_lib.lib().swnerf_sample_pdf(_lib.ptr(bins), _lib.ptr(w), 128, 63, 128, None, _lib.ptr(smp), _lib.ptr(z), 64,
                                            _lib.ptr(zs), _lib.ptr(sd), _lib.stream_of(z)), "sample_pdf")
    zb = float(np.diff(g["bins"], axis=-1).max())
    close_mostly(zs, ref["z_det"], atol=2e-5, frac=0.995, hard=zb + 1e-6, what="sorted union")
    close(sd, ref["std_det"], atol=2e-3, what="z_std")
    assert torch.equal(torch.sort(torch.cat([z, smp], -1), -1)[0], zs)      # the merge itself is exact
    assert bool((zs[:, 1:] >= zs[:, :-1]).all())


@pytest.mark.parametrize("N", [1, 100])
@pytest.mark.parametrize("nb", [2, 51, 501])
@pytest.mark.parametrize("ns", [1, 12, 120])
def test_sample_pdf_size_sweep(sw, dev, N, nb, ns):
    """The size grid of the reference's only unit test for this path (d_nerf/torchsearchsorted/test/test_searchsorted.py:9-44:
    batches {1,100,..} x sorted lengths {1,50,500} x query counts {1,12,120}), applied to the op the search lives in:
    well-conditioned weights (every bin carries mass), random u plus u that hit cdf values exactly (ties go right:
    ray.py:136 `right=True`), against the oracle."""
    rng = np.random.default_rng(1000 * N + 10 * nb + ns)
    bins = np.sort(rng.uniform(2, 6, (N, nb)).astype(np.float32), -1)
    w = rng.uniform(0.5, 1.5, (N, nb - 1)).astype(np.float32)
    u = rng.uniform(0, 1, (N, ns)).astype(np.float32)
    wn = w + np.float32(1e-5)
    cdf = np.concatenate([np.zeros((N, 1), np.float32), np.cumsum(wn / wn.sum(-1, keepdims=True), -1, dtype=np.float32)], -1)
    if ns > 1:
        u[:, 0] = cdf[:, min(1, nb - 1)]                       # exact ties with an interior / the last cdf value
        u[:, -1] = 0.0
    ref = O.sample_pdf(T(bins), T(w), ns, det=False, u=T(u))
    got = sw.ray.sample_pdf(T(bins).to(dev), T(w).to(dev), ns, det=False, u=T(u).to(dev))
    assert got.shape == (N, ns)
    # a tie may resolve to either neighbouring bin when the two float32 cumsums differ in the last bit: both answers
    # lie at the shared bin edge, so the samples still agree
    close(got, ref, atol=5e-5, what=f"sample_pdf N={N} nb={nb} ns={ns}")


# ---------------------------------------------------------------------------------- model.py
def _load(sw, cls, sd_np, dev, **kw):
    m = cls(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, **kw)
    m.load_state_dict({k: T(v) for k, v in sd_np.items()})
    return m.to(dev).eval()


@pytest.fixture(scope="module")
def nets(sw, dev):
    sd_c, sd_f = cases.weights_static()
    e10, _ = sw.embedder.get_embedder(10, 3, 0)
    coarse = _load(sw, sw.model.vallina_NeRF, sd_c, dev)
    fine = _load(sw, sw.model.vallina_NeRF, sd_f, dev)
    orig = _load(sw, sw.model.NeRFOriginal, sd_f, dev, input_ch_time=21, embed_fn=e10)
    dn = sw.model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27,
                                   input_ch_time=21, use_viewdirs=True, embed_fn=e10, zero_canonical=True)
    dn.load_state_dict({k: T(v) for k, v in cases.weights_dnerf().items()})
    return dict(coarse=coarse, fine=fine, orig=orig, dn=dn.to(dev).eval())


def test_state_dict_names_match_reference(sw, nets):
    sd_c, _ = cases.weights_static()
    assert list(nets["coarse"].state_dict().keys()) == list(sd_c.keys())
    assert {k: tuple(v.shape) for k, v in nets["coarse"].state_dict().items()} == {k: v.shape for k, v in sd_c.items()}
    assert set(nets["dn"].state_dict().keys()) == set(cases.weights_dnerf().keys())


def test_mlp_forward_golden(sw, dev, golden, nets):
    g, ref = cases.g4_inputs(), golden("g4_mlp")
    x = T(g["x"]).to(dev)
    # 10 dependent GEMMs with |activations| ~ O(1..10): measured max |delta| ~2e-5 on |out| up to ~30
    close(nets["coarse"](x), ref["vanilla"], atol=1e-4, rtol=1e-4, what="vallina_NeRF")
    out, z = nets["orig"](x, None)
    close(out, ref["original"], atol=1e-4, rtol=1e-4, what="NeRFOriginal")
    assert z.shape == (4096, 3) and float(z.abs().max()) == 0.0
    et, _ = sw.embedder.get_embedder(10, 1, 0)
    for tv in (0.0, 0.5):
        te = et(torch.full((4096, 1), tv, device=dev))
        out, dx = nets["dn"](x, [te, te])
        close(dx, ref[f"dn_dx_t{int(tv*10)}"], atol=2e-5, rtol=1e-4, what=f"dx t={tv}")
        # the canonical net sees gamma(x+dx): a 1e-6 shift in dx is amplified by the 2^9 band
        close(out, ref[f"dn_out_t{int(tv*10)}"], atol=2e-3 if tv else 1e-4, rtol=1e-3, what=f"dnerf out t={tv}")
    # ragged row counts (not a multiple of 32 / of 128), one row, none
    for M in (1, 31, 33, 130):
        close(nets["coarse"](x[:M]), ref["vanilla"][:M], atol=1e-4, rtol=1e-4, what=f"M={M}")
    assert nets["coarse"](x[:0]).shape == (0, 4)
    assert nets["coarse"](x.reshape(64, 64, 90)).shape == (64, 64, 4)


def test_repack_after_weight_update(sw, dev, nets):
    g = cases.g4_inputs()
    x = T(g["x"][:64]).to(dev)
    m = nets["fine"]
    before = m(x).clone()
    with torch.no_grad():
        m.rgb_linear.bias.add_(0.25)              # what an optimizer step does: in-place
    after = m(x)
    np.testing.assert_allclose((after - before)[:, :3].cpu().numpy(), 0.25, atol=1e-5)
    assert float((after - before)[:, 3].abs().max()) == 0.0
    with torch.no_grad():
        m.rgb_linear.bias.sub_(0.25)
    assert torch.allclose(m(x), before, atol=1e-6)


# ------------------------------------------------------------------- render_rays (nerf/run.py)
def _query(sw):
    embed_fn, _ = sw.embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = sw.embedder.get_embedder(4, 3, 0)
    # the lambda create_nerf builds (nerf/run.py:248-251), with the names it closes over
    return lambda inputs, viewdirs, network_fn: sw.render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                      embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)


def _rb(g, dev, t=None):
    return O.make_ray_batch(T(g["rays_o"]), T(g["rays_d"]), g["near"], g["far"], frame_time=t).to(dev)


RGB_TOL = dict(atol=2e-5, rtol=0)          # outputs NOT downstream of resampling (measured ~1e-6 on MI355X)


def bin_width_bound(got, ref, edges):
    """Per-element bound for samples drawn by sample_pdf: a sample whose bin flips across the reference's own
    `denom < 1e-5 -> 1` threshold (ray.py:148-149) stays inside the bin it was drawn in, so |got - ref| <= the
    width of THAT bin (edges [N, nb], sorted; bin of the reference sample), not a global constant."""
    e = np.asarray(edges, np.float64)
    r = np.asarray(ref, np.float64)
    idx = np.stack([np.clip(np.searchsorted(e[i], r[i], side="right") - 1, 0, e.shape[1] - 2) for i in range(e.shape[0])])
    lo = np.take_along_axis(e, idx, 1)
    hi = np.take_along_axis(e, idx + 1, 1)
    return hi - lo


def flips(got, ref, atol=2e-5):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    return int((np.abs(got.astype(np.float64) - ref) > atol).sum())


def z_matched(z_got, z_ref, tol=0.0):
    """For every reference sample (ray i, slot s) the slot of OUR sample of ray i at the SAME depth (bit-equal float32 by
    default), or -1.  Matching is by VALUE, not by slot: one flipped sample shifts the slots of its sorted row.  Why exact:
    raw is a function of gamma(o + d z) whose top band turns one ulp of z (5e-7 at z = 4) into 1e-3 rad of phase and a
    random net's raw moves by up to 5e-2 over that (measured round 3: 5.1e-2 at |dz| <= 1e-6) - only at equal depths is
    the comparison as tight as without resampling.  The 64 coarse depths of a ray are always equal; of the 128 drawn
    ones those whose inverse-cdf arithmetic came out bit-equal."""
    zg = np.asarray(z_got, np.float64)
    zr = np.asarray(z_ref, np.float64)
    idx = np.full(zr.shape, -1, np.int64)
    for i in range(zr.shape[0]):
        j = np.clip(np.searchsorted(zg[i], zr[i]), 0, zg.shape[1] - 1)
        jl = np.clip(j - 1, 0, zg.shape[1] - 1)
        pick = np.where(np.abs(zg[i][jl] - zr[i]) < np.abs(zg[i][j] - zr[i]), jl, j)
        ok = np.abs(zg[i][pick] - zr[i]) <= tol
        idx[i] = np.where(ok, pick, -1)
    return idx


def cmp_per_sample_at_matched_depths(got, ref, z_got, z_ref, what, atol, rtol, min_matched):
    """Per-sample outputs downstream of the resampling (raw, position_delta): a moved sample is a different point, so the
    comparison is made where the depths agree (z_matched) and is then as tight as for a pass without resampling; the
    FRACTION of reference samples that found no partner is bounded separately (these are the flipped samples)."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    z_got = z_got.detach().cpu().numpy() if isinstance(z_got, torch.Tensor) else np.asarray(z_got)
    z_ref = z_ref.detach().cpu().numpy() if isinstance(z_ref, torch.Tensor) else np.asarray(z_ref)
    n = ref.shape[0]
    idx = z_matched(z_got[:n], z_ref[:n])
    m = idx >= 0
    frac = float(m.mean())
    rows = np.broadcast_to(np.arange(n)[:, None], idx.shape)
    g = got[:n][rows[m], idx[m]]
    r = ref[m]
    err = np.abs(g.astype(np.float64) - r)
    lim = atol + rtol * np.abs(r)
    print(f"\n[parity] {what}: {100 * frac:.2f} % of the reference samples have a partner at the same depth; there max |d| = {err.max():.2e} "
          f"(limit {atol:g} + {rtol:g}|ref|), {int((~m).sum())} of {m.size} unmatched")
    assert frac >= min_matched, f"{what}: only {frac:.4f} of the samples matched in depth (need {min_matched})"
    assert np.all(err <= lim), f"{what}: max excess {float((err - lim).max()):.2e} at matched depths"


# Measured on MI355X (round 2, profiles/r02/parity_measured.md): per case and key the fraction of resampled pixels
# within 2e-4 of the reference render runs from 0.937 (S=40/Ni=24 rgb), 0.941 (NDC disp), 0.942 (C2 acc) to 0.992
# (NDC rgb); max |delta| 3.3e-3; PSNR 78.9 (C2), 89.8 (NDC), 81.2 dB (D-NeRF t=0).  Gates: >= 0.92 within 2e-4,
# >= 0.70 within 2e-5 (measured >= 0.77), every element within 2e-2, PSNR floor = measured - 3..4 dB per case (default
# 70 dB = the SURVEY.md 8d floor).  The conditioning argument is in close_mostly's docstring and DESIGN.md 6.
def _cmp(ret, ref, keys, what, nraw=32, resampled=True, scale=1.0, psnr_min=70.0, frac_min=0.92, hard=2e-2, z_pair=None, min_matched=0.5,
         frac_tight=0.70):
    """Compare a render_rays dict with the golden one.  rgb0/disp0/acc0 (and everything when
    N_importance == 0) are held to 2e-5 abs.  Outputs downstream of the hierarchical resampling
    (see close_mostly) are held to: >= 70 % within 2e-5, >= frac_min within 2e-4 (the SURVEY.md 8d
    figure), all within `hard`, and - for the colours - PSNR >= psnr_min dB against the reference render
    (8-bit quantisation noise sits at 58.9 dB).  `scale` widens the two atol bands for D-NeRF with
    t != 0, where gamma(x + dx) multiplies the 2e-7 rounding of dx by 2^9 before the canonical net
    (measured: the ORACLE moves by the same 2e-4 in raw under a +-2e-7 shift of x+dx)."""
    for k in keys:
        v = ret[k]
        w = f"{what}:{k}"
        down = resampled and k in ("rgb_map", "disp_map", "acc_map", "raw", "z_vals", "z_std", "position_delta")
        if k in ("raw", "position_delta"):
            v = v[:nraw]
        if not down:
            if k == "raw":
                close(v, ref[k], atol=1e-3, rtol=1e-4, what=w)                # |raw| up to ~30; D-NeRF: gamma(x+dx) amplifies 2e-7 by 2^9
            elif k.startswith("disp"):
                close(v, ref[k], atol=2e-5, rtol=1e-4, what=w)
            else:
                close(v, ref[k], what=w, **RGB_TOL)
        elif k == "z_vals":
            # sorted union of 64 coarse depths (exact) and 128 drawn samples: an element moves by at most the widest
            # coarse interval of its row when a sample flips (per-row bound from the reference's own z)
            rz = np.asarray(ref[k], np.float64)
            gz = v.detach().cpu().numpy().astype(np.float64)
            # (linear coarse spacing: the union runs from near to far, so one coarse interval = (last - first) / 63)
            row_bound = (rz[:, -1:] - rz[:, :1]) / 63.0 * 1.0001 + 1e-6
            assert np.all(np.abs(gz - rz) <= row_bound), f"{w}: an element moved by more than one coarse interval"
            nfl = flips(gz, rz)
            print(f"\n[parity] {w}: {nfl} of {gz.size} depths differ by > 2e-5 ({100 * nfl / gz.size:.3f} %)")
            close_mostly(v, ref[k], atol=2e-5, frac=0.99, hard=float(row_bound.max()), what=w)
        elif k == "z_std":
            close(v, ref[k], atol=2e-3, what=w)
        elif k in ("raw", "position_delta"):
            # z_pair = (our depths, reference depths) of these rays: compare at equal depths, as tightly as without
            # resampling (raw 1e-3 + 1e-4 |ref|; dx 2e-6) - scaled for D-NeRF t != 0 like the image tolerances
            assert z_pair is not None, f"{w}: per-sample outputs behind the resampling need the depths to be compared at"
            a_, r_ = (1e-3, 1e-4) if k == "raw" else (2e-6, 0.0)
            cmp_per_sample_at_matched_depths(v, ref[k], z_pair[0], z_pair[1], w, a_ * scale, r_ * scale, min_matched)
        else:
            rel = k == "disp_map"
            a_ = v.detach().cpu().numpy()
            d_ = np.abs(np.nan_to_num(a_) - np.nan_to_num(np.asarray(ref[k])))
            if rel:
                d_ = d_ / np.maximum(1.0, np.abs(np.nan_to_num(np.asarray(ref[k]))))
            print(f"\n[parity] {w}: within 2e-5 {float((d_ <= 2e-5 * scale).mean()):.4f}, within 2e-4 {float((d_ <= 2e-4 * scale).mean()):.4f}, "
                  f"max {float(d_.max()):.2e}")
            close_mostly(v, ref[k], atol=2e-5 * scale, frac=frac_tight, hard=hard, what=w, rel=rel)
            close_mostly(v, ref[k], atol=2e-4 * scale, frac=frac_min, hard=hard, what=w, rel=rel)
            if k == "rgb_map" and v.shape[0] >= 128:
                db = psnr(v, ref[k])
                print(f"\n[parity] {w}: PSNR vs reference render {db:.1f} dB, max|d| {maxdiff(v, ref[k]):.2e}")
                assert db >= psnr_min, f"{w}: PSNR {db:.1f} dB < {psnr_min}"


def test_render_rays_static_golden(sw, dev, golden, nets):
    q = _query(sw)
    g = cases.g7_inputs()
    rb = _rb(g, dev)
    assert sw.render.fused_plan(q, [nets["coarse"], nets["fine"]]) == (10, 4, 0)
    r = sw.render.render_rays(rb, nets["coarse"], q, 64, retraw=True, N_importance=0, white_bkgd=True)
    assert list(r.keys()) == ["rgb_map", "disp_map", "acc_map", "raw"]
    _cmp(r, golden("g7_c1"), ["rgb_map", "disp_map", "acc_map", "raw"], "C1", resampled=False)
    r = sw.render.render_rays(rb, nets["coarse"], q, 64, retraw=True, N_importance=128, network_fine=nets["fine"], white_bkgd=True)
    assert list(r.keys()) == ["rgb_map", "disp_map", "acc_map", "raw", "rgb0", "disp0", "acc0", "z_std"]
    assert r["raw"].shape == (1024, 192, 4)
    # the static render_rays does not return its depths (nerf/run.py:405-416): ours from the coarse pass alone, the
    # reference's from golden G7z - captured inside the very call that produced g7_c2 (tests/golden/make_golden_depths.py) -
    # for the 32 rays whose raw the golden holds.  (The oracle's depths will not do: run on 32 rays instead of 1024 its
    # sgemm blocks differently and its samples differ from the reference's in the last bit - measured 5e-2 in raw.)
    z_ours = sw.render.render_pass(rb[:32], nets["coarse"], 64, white_bkgd=True, want=[], n_importance=128)["z_fine"]
    z_ref = golden("g7_c2_depths")["z_vals"]
    _cmp(r, golden("g7_c2"), ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std", "raw"], "C2", psnr_min=75.0,   # measured 78.9
         z_pair=(z_ours, z_ref), min_matched=0.6)
    gs = cases.g7_inputs(n=256, seed=11)
    r = sw.render.render_rays(_rb(gs, dev), nets["coarse"], q, 64, N_importance=128, network_fine=None, white_bkgd=False, lindisp=True)
    _cmp(r, golden("g7_lindisp"), ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"], "lindisp")
    r = sw.render.render_rays(_rb(gs, dev), nets["coarse"], q, 64, N_importance=128, network_fine=nets["fine"], white_bkgd=True,
                              perturb=1., pytest=True)
    _cmp(r, golden("g7_perturb"), ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"], "perturb")


def test_render_ndc_golden(sw, dev, golden, nets):
    gn, ref = cases.g7_ndc_inputs(), golden("g7_ndc")
    Kf, _ = cases.synth.fern_camera()
    rr = sw.render.render(378, 504, Kf, chunk=1024 * 32, rays=(T(gn["rays_o"]).to(dev), T(gn["rays_d"]).to(dev)), ndc=True,
                          near=0., far=1., use_viewdirs=True, network_fn=nets["coarse"], network_query_fn=_query(sw),
                          N_samples=64, N_importance=128, network_fine=nets["fine"], white_bkgd=False, perturb=0., raw_noise_std=0.)
    got = dict(rgb_map=rr[0], disp_map=rr[1], acc_map=rr[2], **rr[3])
    _cmp(got, ref, ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"], "ndc", psnr_min=86.0)   # measured 89.8


def test_c3_ndc_batch_full_size_properties(sw, dev, nets):
    """BASELINE config C3 at its FULL size (fern-like NDC rays, N_rand = 4096, 64+128; the golden G7-ndc holds 256 rays):
    size-independent properties of the render - a sub-batch renders to the same bits as inside the big batch (rays are
    independent), chunked == unchunked, acc in [0, 1], rgb finite and inside [0, 1] (no white background: rgb = sum w.c),
    NaN disparity exactly on the empty rays, and the gates of the golden case against the CPU oracle on 256 rays spread
    evenly over the batch."""
    Kf, c2wf = cases.synth.fern_camera()
    o, d = cases.synth.pick_rays(378, 504, Kf, c2wf, 4096, 3)
    kw = dict(ndc=True, near=0., far=1., use_viewdirs=True, network_fn=nets["coarse"], network_query_fn=_query(sw), N_samples=64,
              N_importance=128, network_fine=nets["fine"], white_bkgd=False, perturb=0., raw_noise_std=0.)
    full = sw.render.render(378, 504, Kf, chunk=1024 * 32, rays=(T(o).to(dev), T(d).to(dev)), **kw)
    assert full[0].shape == (4096, 3) and full[1].shape == (4096,) and full[2].shape == (4096,)
    chunked = sw.render.render(378, 504, Kf, chunk=1000, rays=(T(o).to(dev), T(d).to(dev)), **kw)
    part = sw.render.render(378, 504, Kf, chunk=1024 * 32, rays=(T(o[1500:1757]).to(dev), T(d[1500:1757]).to(dev)), **kw)
    same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))     # disp is NaN on empty rays
    for i in range(3):
        assert same(full[i], chunked[i]) and same(full[i][1500:1757], part[i])
    rgb, disp, acc = (t.cpu().numpy() for t in full[:3])
    assert np.isfinite(rgb).all() and rgb.min() >= -1e-6 and rgb.max() <= 1 + 1e-5
    assert acc.min() >= -1e-6 and acc.max() <= 1 + 1e-5 and 0.01 < acc.mean() < 0.99, (acc.min(), acc.max(), acc.mean())
    assert np.array_equal(np.isnan(disp), acc == 0.0)                       # NaN disparity exactly on empty rays (ray.py:192)
    sel = np.linspace(0, 4095, 256).astype(np.int64)
    sd_c, sd_f = (O.to_torch_sd(s_) for s_ in cases.weights_static())
    rb = O.make_ray_batch(T(o[sel]), T(d[sel]), 0., 1., ndc=True, H=378, W=504, focal=float(Kf[0][0]))
    ref = O.render_rays(rb, sd_c, sd_f, 64, 128, white_bkgd=False)
    got = dict(rgb_map=full[0][sel], disp_map=full[1][sel], acc_map=full[2][sel], **{k: v[sel] for k, v in full[3].items()})
    # (hard bound: NDC disparities run up to ~60; measured max relative |d disp| 2.6e-2 on this subset, 1.7e-2 on the lindisp golden)
    _cmp(got, ref, ["rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"], "C3 full size", psnr_min=86.0, hard=4e-2)


def test_startup_shaping_changes_no_bit(sw, dev, nets, monkeypatch):
    """The start-up shaping of a launch (csrc/render_pass.h pass_startup: L2 warm-up of the weight stream by LDS-DMA into a
    junk slot that later holds the parked encodings, waves starting 0..15 ring steps apart) is timing only: every output of
    the fused passes - static, D-NeRF, without view directions is covered by the soak - is bit-identical with it off."""
    q, qd = _query(sw), _query_d(sw)
    g = cases.g7_inputs(n=1027, seed=77)                       # one full round of waves and a ragged workgroup
    outs = {}
    for warm, skew in (("0", "0"), ("1", "2"), ("1", "1"), ("0", "2")):
        monkeypatch.setenv("SWNERF_WARM", warm)
        monkeypatch.setenv("SWNERF_SKEW", skew)
        a = sw.render.render_rays(_rb(g, dev), nets["coarse"], q, 64, retraw=True, N_importance=128, network_fine=nets["fine"], white_bkgd=True)
        b = sw.render_dnerf.render_rays(_rb(g, dev, 0.5)[:300], nets["dn"], qd, 64, retraw=True, N_importance=128, white_bkgd=True)
        outs[(warm, skew)] = [v.clone() for v in a.values()] + [v.clone() for v in b.values()]
    base = outs[("0", "0")]
    for k, v in outs.items():
        for x, y in zip(base, v):
            assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)), k


def test_render_full_image_c2w_and_chunking(sw, dev, nets):
    """render(c2w=...) on a small frame == oracle; chunked == unchunked bit for bit."""
    K, c2w = cases.synth.lego_camera(24, 40, theta=10.0)
    kw = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets["coarse"], network_query_fn=_query(sw),
              N_samples=64, N_importance=128, network_fine=nets["fine"], white_bkgd=True, perturb=0., raw_noise_std=0.)
    a = sw.render.render(24, 40, K, chunk=1024 * 32, c2w=T(c2w).to(dev), **kw)
    b = sw.render.render(24, 40, K, chunk=100, c2w=T(c2w).to(dev), **kw)
    assert a[0].shape == (24, 40, 3) and a[1].shape == (24, 40)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    o, d = O.get_rays(24, 40, K, c2w)
    sd_c, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    ref = O.render_rays(O.make_ray_batch(o, d, 2., 6.), sd_c, sd_f, 64, 128, white_bkgd=True)
    close_mostly(a[0].reshape(-1, 3), ref["rgb_map"], atol=2e-4, frac=0.92, hard=2e-2, what="render(c2w)")
    close(a[3]["rgb0"].reshape(-1, 3), ref["rgb0"], what="render(c2w) rgb0", **RGB_TOL)


def test_render_path_writes_frames(sw, dev, nets, tmp_path):
    """render_path (nerf/run.py:172-219, d_nerf/run_dnerf.py:175-235): stacked frames == render() per pose, and with
    `savedir` the PNG of every frame decodes to to8b(rgb) (the reference writes them with imageio)."""
    from PIL import Image
    H, W = 20, 28
    poses = [T(cases.synth.lego_camera(H, W, theta=th)[1]).to(dev) for th in (0.0, 40.0)]
    K = cases.synth.lego_camera(H, W)[0]
    kw = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets["coarse"], network_query_fn=_query(sw),
              N_samples=64, N_importance=128, network_fine=nets["fine"], white_bkgd=True, perturb=0., raw_noise_std=0.)
    d = tmp_path / "static"
    d.mkdir()
    rgbs, disps = sw.render.render_path(poses, (H, W, float(K[0, 0])), K, 1024 * 32, kw, savedir=str(d))
    assert rgbs.shape == (2, H, W, 3) and disps.shape == (2, H, W)
    one = sw.render.render(H, W, K, chunk=1024 * 32, c2w=poses[1][:3, :4], **kw)[0]
    assert np.array_equal(rgbs[1], one.cpu().numpy())
    for i in range(2):
        assert np.array_equal(np.asarray(Image.open(str(d / f"{i:03d}.png"))), sw.ray.to8b(rgbs[i]))
    half, _ = sw.render.render_path(poses[:1], (H, W, float(K[0, 0])), K, 1024 * 32, kw, render_factor=2)
    assert half.shape == (1, H // 2, W // 2, 3)
    # D-NeRF: one time per pose, estim/ (+ gt/) sub-directories, i_offset
    kwd = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets["dn"], network_query_fn=_query_d(sw),
               N_samples=64, N_importance=128, white_bkgd=True, perturb=0., raw_noise_std=0.)
    gt = np.random.default_rng(1).uniform(0, 1, (2, H, W, 3)).astype(np.float32)
    r2, _ = sw.render_dnerf.render_path(poses, [0.0, 0.5], (H, W, float(K[0, 0])), 1024 * 32, kwd, gt_imgs=gt,
                                        savedir=str(tmp_path / "dn"), save_also_gt=True, i_offset=7)
    assert r2.shape == (2, H, W, 3)
    for i in range(2):
        assert np.array_equal(np.asarray(Image.open(str(tmp_path / "dn" / "estim" / f"{i + 7:03d}.png"))), sw.ray.to8b(r2[i]))
        assert np.array_equal(np.asarray(Image.open(str(tmp_path / "dn" / "gt" / f"{i + 7:03d}.png"))), sw.ray.to8b(gt[i]))


def test_fused_equals_unfused(sw, dev, nets):
    """The fused pass and the op-by-op path (embed -> mlp_forward -> raw2outputs -> sample_pdf) agree."""
    g = cases.g7_inputs(n=128, seed=21)
    rb = _rb(g, dev)
    q = _query(sw)
    opaque = lambda inputs, viewdirs, network_fn, _q=q: _q(inputs, viewdirs, network_fn)     # hides the encoders
    assert sw.render.fused_plan(opaque, [nets["coarse"]]) is None
    for kw in (dict(N_importance=0), dict(N_importance=128, network_fine=nets["fine"]),
               dict(N_importance=128, network_fine=nets["fine"], perturb=1., pytest=True, raw_noise_std=1.0)):
        a = sw.render.render_rays(rb, nets["coarse"], q, 64, retraw=True, white_bkgd=True, **kw)
        b = sw.render.render_rays(rb, nets["coarse"], opaque, 64, retraw=True, white_bkgd=True, **kw)
        assert list(a.keys()) == list(b.keys())
        if kw["N_importance"] > 0:
            # same arithmetic both ways (1e-7): the depths are compared directly, raw at equal depths.
            # (The fused pass's depths come from its coarse launch; the op path's from the ops it is made of.)
            keys = [k for k in a.keys() if k != "raw"]
        else:
            keys = list(a.keys())
        _cmp(a, {k: v.cpu().numpy() for k, v in b.items()}, keys, f"fused/unfused {list(kw)}", nraw=10**9,
             resampled=kw["N_importance"] > 0)


def test_render_rays_ragged_and_edges(sw, dev, nets):
    """N_samples / N_importance not multiples of 32, N not a multiple of 4, N=1, N=0."""
    q = _query(sw)
    sd_c, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    g = cases.g7_inputs(n=257, seed=33)                       # 257 rays: a last workgroup with one live wave
    for (S, Ni) in ((40, 24), (33, 95), (64, 0), (7, 5), (50, 51), (64, 37), (100, 0)):     # fine pass: 64, 128, -, 12, 101, 101 samples
        r = sw.render.render_rays(_rb(g, dev), nets["coarse"], q, S, N_importance=Ni, network_fine=nets["fine"], white_bkgd=True)
        ref = O.render_rays(_rb(g, "cpu"), sd_c, sd_f, S, Ni, white_bkgd=True)
        # the gate of the 1024-ray golden cases (round 2 ran 37 rays here, where one flipped ray is 2.7 % of the batch)
        # (PSNR: 33 or 40 coarse samples make bins twice as wide as the 64-sample goldens' - a flipped sample moves a pixel
        # further; measured 68.8 dB at S=33 - hence 65 instead of 70)
        _cmp(r, ref, list(r.keys()), f"S={S} Ni={Ni}", resampled=Ni > 0, frac_min=0.92, psnr_min=65.0)
    one = sw.render.render_rays(_rb(g, dev)[:1], nets["coarse"], q, 64, N_importance=128, network_fine=nets["fine"], white_bkgd=True)
    allr = sw.render.render_rays(_rb(g, dev), nets["coarse"], q, 64, N_importance=128, network_fine=nets["fine"], white_bkgd=True)
    assert torch.equal(one["rgb_map"], allr["rgb_map"][:1])
    none = sw.render.render_rays(_rb(g, dev)[:0], nets["coarse"], q, 64, N_importance=128, network_fine=nets["fine"])
    assert none["rgb_map"].shape == (0, 3) and none["z_std"].shape == (0,)


def test_render_rays_beyond_the_lds_slice(sw, dev, nets):
    """The reference takes any N_samples / N_importance (nerf/run.py:361-400).  The fused coarse pass resamples in an LDS
    slice of 256 coarse / 1024 merged depths; beyond that render_rays must NOT fail (round-2 VERDICT: SWNERF_E_UNSUPP
    surfaced as a RuntimeError) but run the fused pass without resampling + sample_pdf / sort as ops + the fused fine pass."""
    q = _query(sw)
    sd_c, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    g = cases.g7_inputs(n=256, seed=41)
    assert not sw.render.pass_can_resample(300, 64) and not sw.render.pass_can_resample(64, 1000) and sw.render.pass_can_resample(256, 768)
    for (S, Ni) in ((300, 64), (64, 1000), (257, 0)):
        r = sw.render.render_rays(_rb(g, dev), nets["coarse"], q, S, N_importance=Ni, network_fine=nets["fine"], white_bkgd=True, retraw=True)
        ref = O.render_rays(_rb(g, "cpu"), sd_c, sd_f, S, Ni, white_bkgd=True, retraw=True)
        assert r["raw"].shape == (256, S + Ni, 4)
        keys = [k for k in r.keys() if k != "raw"]
        # (1000 drawn samples over 63 bins: ten times the draws of the golden cases per bin, so more of them sit at a flipping
        # bin edge - the 2e-5 band is relaxed for that row, the 2e-4 band and the hard bound are not)
        _cmp(r, ref, keys, f"beyond LDS S={S} Ni={Ni}", resampled=Ni > 0, frac_min=0.85, psnr_min=65.0, frac_tight=0.5 if Ni == 1000 else 0.70)
    # the same through the D-NeRF runner's render_rays (t = 0: the canonical net alone, well conditioned)
    qd = _query_d(sw)
    gd = cases.g8_inputs()
    rb = _rb(gd, dev, 0.0)[:128]
    r = sw.render_dnerf.render_rays(rb, nets["dn"], qd, 300, N_importance=64, white_bkgd=True)
    ref = O.render_rays_dnerf(rb.cpu(), O.to_torch_sd(cases.weights_dnerf()), 300, 64, white_bkgd=True)
    assert r["z_vals"].shape == (128, 364)
    close_mostly(r["rgb_map"], ref["rgb_map"], atol=2e-4, frac=0.85, hard=2e-2, what="dnerf beyond LDS rgb")
    close(r["z_std"], ref["z_std"], atol=2e-3, what="dnerf beyond LDS z_std")


# -------------------------------------------------------------- render_rays (d_nerf/run_dnerf.py)
def _query_d(sw):
    embed_fn, _ = sw.embedder.get_embedder(10, 3, 0)
    embeddirs_fn, _ = sw.embedder.get_embedder(4, 3, 0)
    embedtime_fn, _ = sw.embedder.get_embedder(10, 1, 0)
    return lambda inputs, viewdirs, ts, network_fn: sw.render_dnerf.run_network(
        inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn,
        netchunk=1024 * 64, embd_time_discr=True)


def test_render_rays_dnerf_golden(sw, dev, golden, nets):
    qd = _query_d(sw)
    g = cases.g8_inputs()
    assert sw.render.fused_plan(qd, [nets["dn"], None], need_time=True) == (10, 4, 10)
    for tv in (0.0, 0.5):
        ref = golden(f"g8_dnerf_t{int(tv*10)}")
        r = sw.render_dnerf.render_rays(_rb(g, dev, tv), nets["dn"], qd, 64, retraw=True, N_importance=128, white_bkgd=True)
        assert list(r.keys()) == ["rgb_map", "disp_map", "acc_map", "z_vals", "position_delta", "raw", "z_std"]
        if tv == 0.0:
            _cmp(r, ref, ["rgb_map", "disp_map", "acc_map", "z_vals", "z_std", "position_delta", "raw"], f"dnerf t={tv}", psnr_min=78.0,   # measured 81.2
                 z_pair=(r["z_vals"][:32], ref["z_vals"][:32]), min_matched=0.6)      # measured 70.8 % bit-equal depths; there raw 6e-6, dx 0
            continue
        # t != 0: the deformation output dx (ours differs from the reference by <= 2.1e-7, checked in
        # the no-resampling block below) enters gamma(x+dx), whose top band multiplies it by 2^9, BEFORE
        # the coarse weights that drive the resampling.  Calibrate against the reference itself: shift
        # the ORACLE's _time_out.bias by 2e-7 and demand that we are as close to the golden render as
        # that perturbed oracle is (within 6 dB).
        sd = O.to_torch_sd(cases.weights_dnerf())
        sd["_time_out.bias"] = sd["_time_out.bias"] + 2e-7
        pert = O.render_rays_dnerf(_rb(g, "cpu", tv)[:256], sd, 64, 128, white_bkgd=True)
        self_db = psnr(pert["rgb_map"], ref["rgb_map"][:256])
        ours_db = psnr(r["rgb_map"][:256], ref["rgb_map"][:256])
        print(f"\n[parity] dnerf t={tv}: PSNR ours vs reference {ours_db:.1f} dB; reference vs itself under a 2e-7 shift of dx {self_db:.1f} dB")
        # absolute floor (measured 55.3 dB, round 1 and 2) AND the self-calibration (within 3 dB of what a 2e-7 shift does to the reference)
        assert ours_db >= 52.0 and ours_db >= min(65.0, self_db - 3.0)
        drgb = (r["rgb_map"].cpu() - T(ref["rgb_map"])).abs()
        print(f"\n[parity] dnerf t={tv}: rgb within 2e-3 {float((drgb <= 2e-3).float().mean()):.4f}, max {float(drgb.max()):.2e}")
        close_mostly(r["rgb_map"], ref["rgb_map"], atol=2e-3, frac=0.9, hard=0.05, what="dnerf t=0.5 rgb")   # measured below; C5 shard 0.97 within 2e-3, max 2e-2
        # depths: the same per-row bound as the static cases - an element moves by at most one coarse interval of its row
        rz = np.asarray(ref["z_vals"], np.float64)
        gz = r["z_vals"].cpu().numpy().astype(np.float64)
        row_bound = (rz[:, -1:] - rz[:, :1]) / 63.0 * 1.0001 + 1e-6
        nfl = flips(gz, rz)
        print(f"\n[parity] dnerf t={tv}: {nfl} of {gz.size} depths differ by > 2e-5 ({100 * nfl / gz.size:.3f} %), max {np.abs(gz - rz).max():.3e} "
              f"(row bound {row_bound.max():.3f})")
        assert np.all(np.abs(gz - rz) <= row_bound), "dnerf t=0.5 z_vals: an element moved by more than one coarse interval"
        close_mostly(r["z_vals"], ref["z_vals"], atol=2e-5, frac=0.9, hard=float(row_bound.max()), what="dnerf t=0.5 z_vals")
        # per-sample outputs at equal depths - at least the 64 coarse depths of every ray (measured 41.9 %): the limits of the
        # pass without resampling (measured there: dx 1.7e-7, raw 9.3e-5)
        cmp_per_sample_at_matched_depths(r["position_delta"], ref["position_delta"], r["z_vals"], ref["z_vals"], "dnerf t=0.5 position_delta",
                                         atol=2e-6, rtol=0.0, min_matched=0.35)
        cmp_per_sample_at_matched_depths(r["raw"], ref["raw"], r["z_vals"], ref["z_vals"], "dnerf t=0.5 raw", atol=1e-3, rtol=1e-4, min_matched=0.35)
        # the same pass with NO resampling in between is tight
        r0 = sw.render_dnerf.render_rays(_rb(g, dev, tv), nets["dn"], qd, 64, retraw=True, N_importance=0, white_bkgd=True)
        o0 = O.render_rays_dnerf(_rb(g, "cpu", tv)[:128], O.to_torch_sd(cases.weights_dnerf()), 64, 0, white_bkgd=True, retraw=True)
        close(r0["position_delta"][:128], o0["position_delta"], atol=1e-6, what="dx")
        close(r0["rgb_map"][:128], o0["rgb_map"], what="dnerf t=0.5 coarse-only rgb", **RGB_TOL)
        close(r0["raw"][:128], o0["raw"], atol=1e-3, rtol=1e-4, what="dnerf t=0.5 coarse-only raw")
        if tv == 0.0:
            assert float(r["position_delta"].abs().max()) == 0.0
    ref = golden("g8_dnerf_coarse_only")
    r = sw.render_dnerf.render_rays(_rb(g, dev, 0.25)[:128], nets["dn"], qd, 64, N_importance=0, white_bkgd=True)
    _cmp(r, ref, ["rgb_map", "disp_map", "acc_map", "z_vals", "position_delta"], "dnerf coarse only", resampled=False)
    # external z_vals reuse (run_dnerf.py:367,408) and the unfused path
    z = r["z_vals"]
    r2 = sw.render_dnerf.render_rays(_rb(g, dev, 0.25)[:128], nets["dn"], qd, 64, N_importance=0, white_bkgd=True, z_vals=z)
    assert torch.equal(r2["rgb_map"], r["rgb_map"])
    opaque = lambda a, b, c, d, _q=qd: _q(a, b, c, d)
    r3 = sw.render_dnerf.render_rays(_rb(g, dev, 0.5)[:64], nets["dn"], opaque, 64, N_importance=128, white_bkgd=True)
    r4 = sw.render_dnerf.render_rays(_rb(g, dev, 0.5)[:64], nets["dn"], qd, 64, N_importance=128, white_bkgd=True)
    close_mostly(r3["rgb_map"], r4["rgb_map"], atol=2e-4, frac=0.92, hard=2e-2, what="dnerf fused/unfused")
    # without resampling the two paths run the same additions in the same order - layer 0 of the deformation net as bias, then the
    # time terms (once per ray in the fused pass, in line in mlp_forward), then the position terms: raw and dx are the same BITS
    r5 = sw.render_dnerf.render_rays(_rb(g, dev, 0.5)[:96], nets["dn"], opaque, 64, N_importance=0, white_bkgd=True, retraw=True)
    r6 = sw.render_dnerf.render_rays(_rb(g, dev, 0.5)[:96], nets["dn"], qd, 64, N_importance=0, white_bkgd=True, retraw=True)
    assert torch.equal(r5["position_delta"], r6["position_delta"]) and torch.equal(r5["raw"], r6["raw"])
    close(r5["rgb_map"], r6["rgb_map"], atol=1e-6, what="dnerf fused/unfused rgb, no resampling")
    with pytest.raises(AssertionError):
        rb = _rb(g, dev, 0.5)[:8].clone()
        rb[3, 8] = 0.75
        sw.render_dnerf.render_rays(rb, nets["dn"], qd, 64)


# ------------------------------------------------------- mesh grid query
def test_mesh_grid_query_golden(sw, dev, golden, nets):
    import swnerf.mesh as mesh
    ref = golden("g10_mesh_query")
    R = cases.G10_RES
    dens, col, (X, Y, Z) = mesh.sample_grid(cases.G10_BOUNDS, R, nets["fine"], num_views=cases.G10_VIEWS)
    assert dens.shape == (R, R, R) and col.shape == (R, R, R, 3) and dens.dtype == np.float64 and np.array_equal(X, ref["X"])
    # |raw| up to ~10 after 10 fp32 GEMM layers: measured 2e-5
    close(dens, ref["density"], atol=1e-4, rtol=1e-4, what="density field")
    close(col, ref["color"], atol=1e-4, rtol=1e-4, what="colour field")
    pts = T(np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1).astype(np.float32)).to(dev)
    vd = mesh.generate_viewdirs(100)
    d3 = T(np.tile(vd[3][None], (40, 1)).astype(np.float32)).to(dev)
    one = mesh.query_points(nets["fine"], pts[:40], d3)                      # one direction per point
    close(one, ref["per_point_raw"], atol=1e-4, rtol=1e-4, what="per-point raw")
    # the 2-D form of network_query_fn (nerf/load_model.py:56-74) goes through the same kernel
    e10, _ = sw.embedder.get_embedder(10, 3, 0)
    e4, _ = sw.embedder.get_embedder(4, 3, 0)
    two_d = sw.render.run_network(pts[:40], d3, nets["fine"], e10, e4)
    assert two_d.shape == (40, 1, 4) and torch.equal(two_d[:, 0], one)
    # V shared directions == mean of V per-point queries; density is view independent
    dirs = T(mesh.generate_viewdirs(5).astype(np.float32)).to(dev)
    shared = mesh.query_points(nets["fine"], pts, dirs, shared_dirs=True)
    per = torch.stack([mesh.query_points(nets["fine"], pts, dirs[v:v + 1].expand(pts.shape[0], 3).contiguous()) for v in range(5)], 0)
    close(shared[:, :3], per[..., :3].mean(0), atol=1e-5, what="shared == mean of per-point")
    assert torch.equal(shared[:, 3], per[0, :, 3])
    # against the op-by-op path (embed -> cat -> mlp_forward) and ragged sizes
    x = torch.cat([e10(pts[:77]), e4(dirs[2:3].expand(77, 3))], -1)
    close(mesh.query_points(nets["fine"], pts[:77], dirs[2:3].expand(77, 3).contiguous()), nets["fine"](x), atol=2e-6, what="fused == unfused")
    assert mesh.query_points(nets["fine"], pts[:0], dirs, shared_dirs=True).shape == (0, 4)
    with pytest.raises(RuntimeError, match="one per point"):
        mesh.query_points(nets["fine"], pts[:10], dirs, shared_dirs=False)


# ----------------------------------------------------------------- a4: the ray-batch pack, on its own
def test_pack_ray_batch_direct(sw, dev):
    """swnerf_pack_ray_batch (nerf/run.py:137-158, d_nerf/run_dnerf.py:137-160) against the oracle's restatement of the
    same lines, column by column: 11 and 12 columns, ndc on/off (view directions are normalised BEFORE the NDC warp),
    scalar and per-ray near/far, a stride-0 rays_o (what get_rays returns)."""
    Kf, c2wf = cases.synth.fern_camera()
    o, d = cases.synth.pick_rays(378, 504, Kf, c2wf, 777, seed=41)          # N not a multiple of the block size
    rng = np.random.default_rng(42)
    near_a = rng.uniform(0.5, 2.0, (777,)).astype(np.float32)
    far_a = (near_a + rng.uniform(1.0, 4.0, (777,))).astype(np.float32)
    for ndc in (False, True):
        for ft in (None, 0.0, 0.625):
            ref = O.make_ray_batch(T(o), T(d), 2., 6., frame_time=ft, ndc=ndc, H=378, W=504, focal=float(Kf[0][0]))
            got = sw.render.pack_ray_batch(T(o).to(dev), T(d).to(dev), 2., 6., frame_time=ft, ndc=ndc, H=378, W=504, focal=float(Kf[0][0]))
            assert got.shape == ref.shape == (777, 11 if ft is None else 12)
            cols = got.shape[1]
            close(got[:, 6:cols - 3], ref[:, 6:cols - 3], atol=0, what="near/far/(t) columns")
            close(got[:, -3:], ref[:, -3:], atol=1.2e-7, what="viewdirs")                    # d/|d|: 1 ulp of the rsqrt/div chain
            close(got[:, :6], ref[:, :6], atol=2e-7 if ndc else 0, rtol=2e-6 if ndc else 0, what=f"o,d ndc={ndc}")
    # per-ray near/far arrays (render()'s docstring allows them), 2-D [H,W,3] inputs, stride-0 origin
    ref = O.make_ray_batch(T(o), T(d), T(near_a)[:, None], T(far_a)[:, None])
    got = sw.render.pack_ray_batch(T(o).to(dev), T(d).to(dev), T(near_a).to(dev), T(far_a).to(dev))
    close(got[:, 6], ref[:, 6], atol=0, what="near array")
    close(got[:, 7], ref[:, 7], atol=0, what="far array")
    centre = T(o[:1]).to(dev).expand(21, 37, 3)
    got = sw.render.pack_ray_batch(centre, T(d[:777].reshape(21, 37, 3)).to(dev), 2., 6.)
    close(got[:, :3], np.broadcast_to(o[:1], (777, 3)), atol=0, what="stride-0 rays_o")
    assert sw.render.pack_ray_batch(T(o[:0]).to(dev), T(d[:0]).to(dev), 2., 6.).shape == (0, 11)


def test_render_rays_dnerf_with_original_net(sw, dev, nets):
    """run_dnerf.py with nerf_type='original' (model.py:214-225, 227-296): the canonical net alone behind the D-NeRF
    render_rays; position_delta is ZEROS (NeRFOriginal.forward returns torch.zeros_like(input_pts[:, :3])), the
    colours equal the static render of the same weights."""
    qd = _query_d(sw)
    g = cases.g8_inputs()
    rb = _rb(g, dev, 0.5)[:200]
    assert sw.render.fused_plan(qd, [nets["orig"], None], need_time=True) == (10, 4, 10)
    for Ni in (0, 128):
        r = sw.render_dnerf.render_rays(rb, nets["orig"], qd, 64, retraw=True, N_importance=Ni, white_bkgd=True,
                                        use_two_models_for_fine=False)
        S = 64 + Ni
        assert r["position_delta"].shape == (200, S, 3) and float(r["position_delta"].abs().max()) == 0.0
        ref = O.render_rays(_rb(g, "cpu")[:200], O.to_torch_sd(cases.weights_static()[1]), None, 64, Ni, white_bkgd=True)
        if Ni == 0:
            close(r["rgb_map"], ref["rgb_map"], what="original-in-dnerf rgb", **RGB_TOL)
        else:
            close_mostly(r["rgb_map"], ref["rgb_map"], atol=2e-4, frac=0.9, hard=5e-2, what="original-in-dnerf rgb (resampled)")
    r2 = sw.render_dnerf.render_rays(rb, nets["orig"], qd, 64, N_importance=128, white_bkgd=True, use_two_models_for_fine=True,
                                     network_fine=nets["orig"])
    assert float(r2["position_delta_0"].abs().max()) == 0.0 and r2["position_delta_0"].shape == (200, 64, 3)


def test_rng_paths_statistics(sw, dev, nets):
    """perturb=1 / raw_noise_std>0 WITHOUT injected randoms (torch's generator; the reference draws torch.rand /
    torch.randn at the same three places: nerf/run.py:375-377, ray.py:117-121, :176-178).  Not bit-comparable, so:
    moments + a Kolmogorov-Smirnov distance of the stratified jitter and of the inverse-CDF samples, and the
    Monte-Carlo mean of the render against the deterministic one."""
    torch.manual_seed(1234)
    g = cases.g7_inputs(n=512, seed=77)
    rb = _rb(g, dev)
    q = _query(sw)
    # (1) stratified jitter (nerf/run.py:369-383): z = lower + (upper-lower) * U[0,1) per (ray, sample)
    t_rand, u, noise = sw.render._rng_inputs(512, 64, 128, 1., 1., False, dev)
    z = sw.render.sample_coarse(rb, 64, False, t_rand)
    z0 = sw.render.sample_coarse(rb, 64, False, None)
    mids = .5 * (z0[:, 1:] + z0[:, :-1])
    upper, lower = torch.cat([mids, z0[:, -1:]], -1), torch.cat([z0[:, :1], mids], -1)
    assert bool((z >= lower - 1e-6).all()) and bool((z <= upper + 1e-6).all())
    frac = ((z - lower) / (upper - lower))[:, 1:-1].reshape(-1).double().cpu().numpy()       # should be U[0,1)
    n = frac.size
    ks = float(np.abs(np.sort(frac) - (np.arange(n) + 0.5) / n).max())
    assert ks < 1.63 / np.sqrt(n) * 1.5, f"stratified jitter: KS distance {ks:.4f} (n={n})"   # 1 % critical value x 1.5
    assert abs(frac.mean() - 0.5) < 4 * np.sqrt(1 / 12 / n) and abs(frac.var() - 1 / 12) < 0.002
    nz = noise(64).double().cpu().numpy().reshape(-1)
    assert abs(nz.mean()) < 4 / np.sqrt(nz.size) and abs(nz.std() - 1.0) < 0.02               # randn * raw_noise_std
    # (2) inverse-CDF samples with random u: cdf(sample) must be U[0,1) again (probability integral transform)
    p0 = sw.render.render_pass(rb, nets["coarse"], 64, white_bkgd=True, want=["weights", "z_out"])
    w, zc = p0["weights"], p0["z_out"]
    bins = .5 * (zc[:, 1:] + zc[:, :-1])
    smp = sw.ray.sample_pdf(bins, w[:, 1:-1], 128, det=False)
    wn = w[:, 1:-1].double().cpu().numpy() + 1e-5
    cdf = np.concatenate([np.zeros((512, 1)), np.cumsum(wn / wn.sum(-1, keepdims=True), -1)], -1)
    back = np.stack([np.interp(smp[r].double().cpu().numpy(), bins[r].double().cpu().numpy(), cdf[r]) for r in range(512)])
    # rows whose mass sits in bins of < 1e-5 snap to bin edges (ray.py:148-149) - the transform is only uniform where
    # the pdf is positive, so test rows that spread their mass
    spread = (wn / wn.sum(-1, keepdims=True)).max(-1) < 0.5
    b = np.sort(back[spread].reshape(-1))
    ks2 = float(np.abs(b - (np.arange(b.size) + 0.5) / b.size).max())
    assert spread.sum() >= 32 and ks2 < 0.02, f"sample_pdf random u: KS distance {ks2:.4f} over {spread.sum()} rows"
    # (3) end to end: the mean of 16 perturbed renders approaches the deterministic render (the estimator is
    # consistent; a stuck or mis-scaled generator shows up as a bias)
    kw = dict(N_importance=128, network_fine=nets["fine"], white_bkgd=True)
    det = sw.render.render_rays(rb, nets["coarse"], q, 64, perturb=0., **kw)["rgb_map"]
    runs = torch.stack([sw.render.render_rays(rb, nets["coarse"], q, 64, perturb=1., **kw)["rgb_map"] for _ in range(16)], 0)
    assert float((runs[0] - runs[1]).abs().max()) > 1e-4                                       # the draws differ
    # With random (untrained) nets whose field carries a 2^9 band, a jittered sample set renders a DIFFERENT colour per
    # pixel (per-pixel std ~0.1), so pixels do not converge to the deterministic render; what must agree is the
    # population statistic: the mean colour over all pixels and runs vs the deterministic frame's mean colour.
    spread_px = runs.std(0).mean()
    bias = (runs.mean(0) - det).mean(0).abs().max()
    print(f"\n[parity] rng: KS jitter {ks:.4f}, KS inverse-cdf {ks2:.4f}, per-pixel std {float(spread_px):.4f}, "
          f"|mean colour of 16 jittered frames - deterministic frame| {float(bias):.4f}")
    assert float(bias) < 0.03 and 0.01 < float(spread_px) < 0.3
    noisy = sw.render.render_rays(rb, nets["coarse"], q, 64, perturb=0., raw_noise_std=1.0, **kw)["rgb_map"]
    assert float((noisy - det).abs().max()) > 1e-5 and bool(torch.isfinite(noisy).all())


# ------------------------------------------------------------------ full-size properties + report
def test_c2_full_size_properties(sw, dev, nets):
    """BASELINE config C2 (lego 800x800, N_rand=4096, 64+128): size-independent properties, plus the
    oracle on a 256-ray subset, plus PSNR of the GPU render against the oracle render."""
    K, c2w = cases.synth.lego_camera(800, 800)
    o, d = cases.synth.pick_rays(800, 800, K, c2w, 4096, seed=2)
    rb = O.make_ray_batch(T(o), T(d), 2., 6.).to(dev)
    q = _query(sw)
    p0 = sw.render.render_pass(rb, nets["coarse"], 64, white_bkgd=True, want=["rgb_map", "acc_map", "weights", "z_out"], n_importance=128)
    zf = p0["z_fine"]
    assert zf.shape == (4096, 192) and bool((zf[:, 1:] >= zf[:, :-1]).all())                  # sortedness
    assert bool(((zf >= 2.0 - 1e-6) & (zf <= 6.0 + 1e-6)).all())
    # every coarse depth is present in the sorted union
    merged = torch.sort(torch.cat([zf, p0["z_out"]], -1), -1)[0]
    assert int((merged[:, 1:] == merged[:, :-1]).sum(-1).min()) >= 64
    close(p0["weights"].sum(-1), p0["acc_map"], atol=2e-6, what="sum(weights) == acc")
    # the merge behind the resampling (render_kernels.hip): random u (unsorted samples) and UNSORTED depths handed in
    # both take the sort-first branch; the result is still the sorted union containing every input depth
    g_ = torch.Generator(device="cpu").manual_seed(5)
    u_r = torch.rand((512, 128), generator=g_).to(dev)
    z_sh = p0["z_out"][:512][:, torch.randperm(64, generator=g_).to(dev)].contiguous()
    for zin, uin in ((None, u_r), (z_sh, None), (z_sh, u_r)):
        pm = sw.render.render_pass(rb[:512], nets["coarse"], 64, z_vals=zin, white_bkgd=True, want=["z_out"], n_importance=128, u=uin)
        zm = pm["z_fine"]
        assert bool((zm[:, 1:] >= zm[:, :-1]).all())
        both = torch.sort(torch.cat([zm, pm["z_out"]], -1), -1)[0]
        assert int((both[:, 1:] == both[:, :-1]).sum(-1).min()) >= 64
    pm0 = sw.render.render_pass(rb[:512], nets["coarse"], 64, white_bkgd=True, want=["weights", "z_out"], n_importance=128, u=u_r)
    bins_ = .5 * (pm0["z_out"][:, 1:] + pm0["z_out"][:, :-1])
    smp_ = sw.ray.sample_pdf(bins_, pm0["weights"][:, 1:-1], 128, u=u_r)
    assert torch.equal(pm0["z_fine"], torch.sort(torch.cat([pm0["z_out"], smp_], -1), -1)[0])     # == torch.sort of the op path
    assert bool((p0["weights"] >= 0).all()) and float(p0["acc_map"].max()) <= 1.0 + 1e-5
    p1 = sw.render.render_pass(rb, nets["fine"], 192, z_vals=zf, white_bkgd=True, want=["rgb_map", "acc_map", "weights"])
    close(p1["weights"].sum(-1), p1["acc_map"], atol=2e-6, what="sum(weights) == acc (fine)")
    # linearity of compositing in the background term: white - black = 1 - acc
    p1b = sw.render.render_pass(rb, nets["fine"], 192, z_vals=zf, white_bkgd=False, want=["rgb_map"])
    close(p1["rgb_map"] - p1b["rgb_map"], (1 - p1["acc_map"])[:, None].expand(-1, 3), atol=1e-6, what="white-bkgd identity")
    # idempotence / determinism
    again = sw.render.render_rays(rb, nets["coarse"], q, 64, N_importance=128, network_fine=nets["fine"], white_bkgd=True)
    assert torch.equal(again["rgb_map"], p1["rgb_map"])
    sd_c, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    ref = O.render_rays(rb[:256].cpu(), sd_c, sd_f, 64, 128, white_bkgd=True)
    close_mostly(again["rgb_map"][:256], ref["rgb_map"], atol=2e-4, frac=0.92, hard=2e-2, what="C2 subset")
    close(again["rgb0"][:256], ref["rgb0"], what="C2 subset rgb0", **RGB_TOL)
    # stage-isolated: the fine pass on the ORACLE's own depths (no resampling in between)
    iso = sw.render.render_pass(rb[:256], nets["fine"], 192, z_vals=ref["z_vals"].to(dev), white_bkgd=True, want=["rgb_map", "acc_map"])
    close(iso["rgb_map"], ref["rgb_map"], what="fine pass at the oracle's z", **RGB_TOL)
    db = psnr(again["rgb_map"][:256], ref["rgb_map"])
    print(f"\n[parity] C2: fine pass @oracle z max|d rgb| = {maxdiff(iso['rgb_map'], ref['rgb_map']):.3e}; "
          f"end-to-end max|d rgb| = {maxdiff(again['rgb_map'][:256], ref['rgb_map']):.3e}, PSNR vs oracle render = {db:.1f} dB")
    assert db >= 70.0                                                                         # SURVEY.md 8d floor


def test_errors_are_loud(sw, dev, nets):
    g = cases.g4_inputs()
    with pytest.raises(RuntimeError, match="GPU"):
        nets["coarse"](T(g["x"][:8]))                      # CPU tensor: no CPU path
    with pytest.raises(RuntimeError, match="GPU"):
        sw.ray.raw2outputs(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))
    with pytest.raises(ValueError):
        sw.ray.sample_pdf(torch.zeros(2, 8, device=dev), torch.zeros(2, 8, device=dev), 4)
    with pytest.raises(NotImplementedError):
        sw.embedder.Embedder(include_input=False, input_dims=3, max_freq_log2=9, num_freqs=10, log_sampling=True,
                             periodic_fns=[torch.sin, torch.cos])
    other = sw.model.vallina_NeRF(D=4, W=128, input_ch=63, input_ch_views=27, skips=[2], use_viewdirs=True).to(dev)
    assert other(torch.zeros(4, 90, device=dev)).shape == (4, 4)        # not an error: the generic layer-by-layer path (test_gpu_generic.py)
    with pytest.raises(NotImplementedError):
        other.packed()                                               # ... but no packed stream for the fused kernels
    from swnerf import _lib
    a = _lib.PassArgs()
    rc = _lib.lib().swnerf_render_pass(a, None)
    assert rc == -1 and b"NULL" in _lib.lib().swnerf_last_error()
    with torch.enable_grad():
        # gradients w.r.t. the embedded inputs are not built (rays are data in train()): loud, not silently zero
        x = T(g["x"][:8]).to(dev).requires_grad_(True)
        y = nets["fine"](x)
        with pytest.raises(NotImplementedError, match="embedded inputs"):
            y.sum().backward()
        for p_ in nets["fine"].parameters():
            p_.grad = None
