"""BASELINE configs C4 / C5 on the HIP path at their per-GPU shard sizes: rank r of 8 generates
the rays of its contiguous row-major range of the frame, renders them through the fused pass (64+128) and returns
[rgb, disp, acc] - `swnerf.parallel.frame_renderer`, the function bench.py --config C4|C5 times on every rank and
`render_image_sharded` gathers.  Full-size checks are size-independent properties (determinism, chunk invariance,
shard == the same rows of the whole frame, value ranges); the oracle renders a 256-ray subset.
Reference entries: nerf/run.py:557-571 (render_only), d_nerf/run_dnerf.py:553-566."""
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


def _psnr(a, b):
    return float(-10 * np.log10(max(float(((a - b) ** 2).mean()), 1e-20)))


def _static_scene(dev):
    from swnerf import model, embedder, render
    embed_fn, c10 = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, c4 = embedder.get_embedder(4, 3, 0)
    nets = []
    for sd in cases.weights_static():
        m = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict({k: T(v) for k, v in sd.items()})
        nets.append(m.to(dev).eval())
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    return nets, dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets[0], network_query_fn=q, N_samples=64,
                      N_importance=128, network_fine=nets[1], white_bkgd=True, perturb=0., raw_noise_std=0.)


def test_c4_shard_80000_rays(dev):
    """C4: lego 800x800 render_only over 8 GPUs -> rank 3 renders rays [240000, 320000)."""
    from swnerf import parallel, synth
    H = W = 800
    K, c2w = synth.lego_camera(H, W)
    nets, kw = _static_scene(dev)
    rr = parallel.frame_renderer(H, W, K, c2w, kw, device=dev)
    lo, hi = synth.shard_range(H * W, 8, 3)
    assert (lo, hi) == (240000, 320000)
    px = rr(lo, hi - lo)
    assert px.shape == (80000, 5)
    rgb, disp, acc = px[:, :3], px[:, 3], px[:, 4]
    assert bool(torch.isfinite(rgb).all()) and float(rgb.min()) >= -1e-6 and float(rgb.max()) <= 1.0 + 1e-5
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    assert bool((torch.isnan(disp) == (acc == 0)).all())                     # ray.py:192: NaN exactly for empty rays
    assert 0.05 < float(acc.mean()) < 0.95                                   # the synthetic scene is neither empty nor opaque
    # determinism, and the launch size does not matter (rays are independent): two half shards == the shard, bit for bit
    assert torch.equal(rr(lo, hi - lo), px)
    half = (hi - lo) // 2
    both = torch.cat([rr(lo, half), rr(lo + half, hi - lo - half)], 0)
    assert torch.equal(both.nan_to_num(-1.0), px.nan_to_num(-1.0))
    # the option to render the shard as concurrent sub-ranges on side streams (so that one sub-range's next launch fills the
    # CUs the other's ragged last round leaves idle): two streams, three streams: the same bits
    for ns in (2, 3):
        assert torch.equal(parallel.frame_renderer(H, W, K, c2w, kw, device=dev, streams=ns)(lo, hi - lo).nan_to_num(-1.0), px.nan_to_num(-1.0))
    # the reference's chunk (utils.py:33, 32768 rays) through batchify_rays gives the same pixels
    rr_chunked = parallel.frame_renderer(H, W, K, c2w, kw, chunk=1024 * 32, device=dev)
    assert torch.equal(rr_chunked(lo, hi - lo).nan_to_num(-1.0), px.nan_to_num(-1.0))
    # world = 1 through render_image_sharded: the whole frame; the shard is rows 300..399 of it
    img = parallel.render_image_sharded(rr, H, W)
    assert img.shape == (H, W, 5)
    assert torch.equal(img.reshape(-1, 5)[lo:hi].nan_to_num(-1.0), px.nan_to_num(-1.0))
    # oracle on 256 rays spread over the shard
    sel = np.linspace(0, hi - lo - 1, 256).astype(np.int64)
    o, d = O.get_rays(H, W, K, c2w)
    rb = O.make_ray_batch(o.reshape(-1, 3)[lo + sel], d.reshape(-1, 3)[lo + sel], 2., 6.)
    sd_c, sd_f = (O.to_torch_sd(s) for s in cases.weights_static())
    with torch.no_grad():
        ref = O.render_rays(rb, sd_c, sd_f, 64, 128, white_bkgd=True)
    got = px[T(sel).to(dev)].cpu()
    dlt = (got[:, :3] - ref["rgb_map"]).abs()
    db = _psnr(got[:, :3].numpy(), ref["rgb_map"].numpy())
    print(f"\n[parity] C4 shard (80 000 rays, HIP) vs oracle on 256 of them: within 2e-4 {float((dlt <= 2e-4).float().mean()):.4f}, "
          f"max {float(dlt.max()):.2e}, PSNR {db:.1f} dB")
    assert float((dlt <= 2e-4).float().mean()) >= 0.92 and db >= 72.0       # measured 0.960 / 75.0 dB; SURVEY.md 8d floor is 70
    assert np.array_equal(np.isnan(got[:, 3].numpy()), np.isnan(ref["disp_map"].numpy()))


def test_c5_shard_20000_rays_dnerf(dev):
    """C5: D-NeRF 400x400 frame at t = 0.5 over 8 GPUs -> rank 3 renders rays [60000, 80000) with the
    deformation + canonical net per sample; and the t = 0 frame (zero_canonical branch, model.py:143-145)."""
    from swnerf import parallel, synth, model, embedder, render_dnerf
    H = W = 400
    K, c2w = synth.lego_camera(H, W)
    embed_fn, c10 = embedder.get_embedder(10, 3, 0)
    embeddirs_fn, c4 = embedder.get_embedder(4, 3, 0)
    embedtime_fn, ct = embedder.get_embedder(10, 1, 0)
    dn = model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=c10, output_ch=5, skips=[4], input_ch_views=c4,
                                input_ch_time=ct, use_viewdirs=True, embed_fn=embed_fn, zero_canonical=True)
    dn.load_state_dict({k: T(v) for k, v in cases.weights_dnerf().items()})
    dn = dn.to(dev).eval()
    qd = lambda inputs, viewdirs, ts, network_fn: render_dnerf.run_network(
        inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn,
        netchunk=1024 * 64, embd_time_discr=True)
    kw = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=dn, network_query_fn=qd, N_samples=64,
              N_importance=128, network_fine=None, white_bkgd=True, perturb=0., raw_noise_std=0.)
    lo, hi = synth.shard_range(H * W, 8, 3)
    assert (lo, hi) == (60000, 80000)
    sd = O.to_torch_sd(cases.weights_dnerf())
    o, d = O.get_rays(H, W, float(K[0][0]), c2w)
    sel = np.linspace(0, hi - lo - 1, 256).astype(np.int64)
    for tv, floor_db in ((0.5, 54.0), (0.0, 77.0)):                        # measured 58.5 / 80.4 dB
        rr = parallel.frame_renderer(H, W, K, c2w, kw, frame_time=tv, device=dev)
        px = rr(lo, hi - lo)
        assert px.shape == (20000, 5) and bool(torch.isfinite(px[:, :3]).all())
        assert float(px[:, 4].min()) >= 0.0 and float(px[:, 4].max()) <= 1.0 + 1e-5
        assert torch.equal(rr(lo, hi - lo).nan_to_num(-1.0), px.nan_to_num(-1.0))
        both = torch.cat([rr(lo, 7000), rr(lo + 7000, hi - lo - 7000)], 0)
        assert torch.equal(both.nan_to_num(-1.0), px.nan_to_num(-1.0))
        two_streams = parallel.frame_renderer(H, W, K, c2w, kw, frame_time=tv, device=dev, streams=2)(lo, hi - lo)
        assert torch.equal(two_streams.nan_to_num(-1.0), px.nan_to_num(-1.0))      # two sub-ranges rendered concurrently: the same bits
        rb = O.make_ray_batch(o.reshape(-1, 3)[lo + sel], d.reshape(-1, 3)[lo + sel], 2., 6., frame_time=tv)
        with torch.no_grad():
            ref = O.render_rays_dnerf(rb, sd, 64, 128, white_bkgd=True)
        got = px[T(sel).to(dev)].cpu()
        db = _psnr(got[:, :3].numpy(), ref["rgb_map"].numpy())
        dlt = (got[:, :3] - ref["rgb_map"]).abs()
        print(f"\n[parity] C5 shard t={tv} (20 000 rays, HIP) vs oracle on 256 of them: within 2e-4 {float((dlt <= 2e-4).float().mean()):.4f}, "
              f"within 2e-3 {float((dlt <= 2e-3).float().mean()):.4f}, max {float(dlt.max()):.2e}, PSNR {db:.1f} dB")
        # t = 0.5: gamma(x+dx) amplifies the 2e-7 rounding of dx by 2^9 before the resampling (tests/test_gpu_parity.py
        # test_render_rays_dnerf_golden calibrates this against the reference itself: 54.5 dB under a 2e-7 shift)
        assert db >= floor_db
    # the frame shards tile the image: ranks 3 and 4 of 8 are adjacent rows of the world-1 render
    rr = parallel.frame_renderer(H, W, K, c2w, kw, frame_time=0.5, device=dev)
    lo4, hi4 = synth.shard_range(H * W, 8, 4)
    two = rr(lo, hi4 - lo)
    assert torch.equal(two[:hi - lo].nan_to_num(-1.0), rr(lo, hi - lo).nan_to_num(-1.0))
