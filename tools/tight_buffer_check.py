#!/usr/bin/env python3
"""A poor man's address sanitizer for the C-ABI entry points (GPU AddressSanitizer is not available on this pool): every
operand and output is a torch allocation of at least 10 MB whose size is a multiple of 2 MB - the caching allocator then
gives it a segment of exactly that size, so a kernel that reads or writes one element past its last row leaves the mapping
and faults instead of silently touching a neighbour.  (This is how round 3 found the grid-decode bug of the narrow weight-
gradient GEMM.)  tight_buffer_check.py <case> [<case> ...] runs the cases one after the other in this process (the caching
allocator is emptied in between, so every case gets fresh exact-size segments); `list` prints the cases.  A GPU memory fault
aborts the process (exit code 134 / -6): tests/test_00_a_tight_buffers.py starts it as a child and fails on a non-zero exit OR a
"Memory access fault" line in its output.
Round 4: the generic path's cases - the shape that faulted in round 3 (views_linears.0 of a W = 256 net with view directions
and skips [2, 5]: a 128 x 283 weight gradient) forward + backward, and a D != 8 net."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
CASES = ["rays", "embed", "pass_static", "pass_dnerf", "pass_noview", "raw2outputs", "sample_pdf", "mlp_static", "mlp_dnerf", "mlp_noview",
         "query", "sample_coarse", "train_static", "train_dnerf", "train_noview", "pass_x3_static", "pass_x3_dnerf",
         "generic_w256_views", "generic_d6"]
if len(sys.argv) < 2 or sys.argv[1] == "list":
    print(" ".join(CASES))
    sys.exit(0)
if len(sys.argv) > 2:                                # several cases: each in turn, fresh allocator segments for every one
    import runpy
    import torch as _t
    for c in sys.argv[1:]:
        if c not in CASES:
            raise SystemExit(f"unknown case {c!r}; `list` prints them")
        sys.argv = [sys.argv[0], c]
        runpy.run_path(os.path.abspath(__file__), run_name="__main__")
        _t.cuda.synchronize()
        _t.cuda.empty_cache()
    sys.exit(0)
if sys.argv[1] not in CASES:
    raise SystemExit(f"unknown case {sys.argv[1]!r}; `list` prints them")
import numpy as np
import torch
from swnerf import synth, model, embedder, render, render_dnerf, ray, mesh
render.set_precision("fp32")                     # (several cases may share this process: the x3 cases switch it)

case = sys.argv[1]
dev = torch.device("cuda:0")
N = 1 << 19                                     # 524 288 rays: [N,11] = 11 x 2 MB, [N,12] = 12 x 2 MB, [N,8] = 8 x 2 MB
embed_fn, c10 = embedder.get_embedder(10, 3, 0)
embeddirs_fn, c4 = embedder.get_embedder(4, 3, 0)
embedtime_fn, ct = embedder.get_embedder(10, 1, 0)


def tight(t):
    assert t.numel() * 4 >= 10 << 20 and (t.numel() * 4) % (2 << 20) == 0, (tuple(t.shape), "not a tight allocation")
    return t


def static_net(seed=1):
    m = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(seed, alpha_bias=-0.5).items()})
    return m.to(dev).eval()


def dnerf_net():
    m = model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=c10, output_ch=5, skips=[4], input_ch_views=c4, input_ch_time=ct,
                               use_viewdirs=True, embed_fn=embed_fn, zero_canonical=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.dnerf_state_dict(3, alpha_bias=-1.0).items()})
    return m.to(dev).eval()


def noview_net():
    m = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.noview_state_dict(5, alpha_bias=0.5).items()})
    return m.to(dev).eval()


def ray_batch(cols):
    rb = tight(torch.empty((N, cols), device=dev))
    rb[:, 0:3] = torch.tensor([0.3, -0.2, 4.0], device=dev)
    d = torch.randn((N, 3), device=dev) * 0.1 + torch.tensor([0., 0., -1.], device=dev)
    rb[:, 3:6] = d
    rb[:, 6], rb[:, 7] = 2., 6.
    if cols == 12:
        rb[:, 8] = 0.5
    if cols >= 11:
        rb[:, -3:] = d / d.norm(dim=-1, keepdim=True)
    return rb


with torch.no_grad():
    if case == "rays":
        K, c2w = synth.lego_camera(1024, 512)
        o, d = ray.get_rays_range(1024, 512, K, torch.from_numpy(c2w).to(dev), 0, N, dev)
        o2, d2 = ray.ndc_rays(1024, 512, float(K[0][0]), 1., tight(torch.randn((4 * N, 3), device=dev)), tight(torch.randn((4 * N, 3), device=dev) - 2))
        rb = render.pack_ray_batch(tight(torch.randn((4 * N, 3), device=dev)), tight(torch.randn((4 * N, 3), device=dev)), 2., 6.)
        assert rb.shape == (4 * N, 11) and bool(torch.isfinite(rb).all())
    elif case == "embed":
        x = tight(torch.randn((4 * N, 3), device=dev))
        out = embed_fn(x)
        tight(out)
        assert out.shape == (4 * N, 63) and bool(torch.isfinite(out).all())
        t = embedtime_fn(tight(torch.rand((8 * N, 1), device=dev)))
        assert t.shape == (8 * N, 21)
    elif case in ("pass_static", "pass_noview", "pass_dnerf", "pass_x3_static", "pass_x3_dnerf"):
        if "x3" in case:
            render.set_precision("bf16x3")                   # the opt-in bf16x3 pass (csrc/mlp_core_x3.h)
            case = case.replace("_x3", "")
        net = {"pass_static": static_net, "pass_noview": noview_net, "pass_dnerf": dnerf_net}[case]()
        rb = ray_batch({"pass_static": 11, "pass_noview": 8, "pass_dnerf": 12}[case])
        n = N
        want = ["rgb_map", "disp_map", "acc_map", "depth_map", "weights", "raw", "z_out"] + (["dx"] if case == "pass_dnerf" else [])
        p0 = render.render_pass(rb, net, 64, white_bkgd=True, want=want, n_importance=128, t_rand=tight(torch.rand((n, 64), device=dev)),
                                noise=tight(torch.randn((n, 64), device=dev)), u=tight(torch.rand((n, 128), device=dev)))
        p1 = render.render_pass(rb, net, 192, z_vals=tight(p0["z_fine"]), white_bkgd=True, want=want, noise=tight(torch.randn((n, 192), device=dev)))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(p1["rgb_map"]).all()) and bool((p0["z_fine"][:, 1:] >= p0["z_fine"][:, :-1]).all())
    elif case == "raw2outputs":
        n = 1 << 16
        raw = tight(torch.randn((n, 192, 4), device=dev)).requires_grad_(True)
        z = tight(torch.sort(torch.rand((n, 192), device=dev) * 4 + 2, -1)[0])
        d = torch.randn((n, 3), device=dev)
        with torch.enable_grad():
            out = ray.raw2outputs(raw, z, d, 0., True, noise=tight(torch.randn((n, 192), device=dev)))
            (out[0].sum() + out[2].sum() + (out[3] * out[3]).sum() + out[4].sum()).backward()
        assert bool(torch.isfinite(raw.grad).all())
    elif case == "sample_pdf":
        bins = tight(torch.sort(torch.rand((N, 63), device=dev), -1)[0])
        w = tight(torch.rand((N, 62), device=dev))
        s = ray.sample_pdf(bins, w, 128, det=True)
        s2 = ray.sample_pdf(bins, w, 128, det=False, u=tight(torch.rand((N, 128), device=dev)))
        assert tuple(s.shape) == (N, 128) and bool(torch.isfinite(s2).all())
    elif case in ("mlp_static", "mlp_dnerf", "mlp_noview"):
        M = 1 << 20
        if case == "mlp_noview":
            out = noview_net()(tight(torch.randn((1 << 21, 63), device=dev)))
            assert out.shape[1] == 5
        else:
            x = tight(torch.randn((M, 90), device=dev))
            if case == "mlp_static":
                out = static_net()(x)
                assert out.shape == (M, 4)
            else:
                te = embedtime_fn(torch.full((M, 1), 0.5, device=dev))
                out, dx = dnerf_net()(x, [te, te])
                assert out.shape == (M, 4) and dx.shape == (M, 3)
        assert bool(torch.isfinite(out).all())
    elif case == "query":
        pts = tight(torch.randn((4 * N, 3), device=dev))
        dirs = torch.randn((100, 3), device=dev)
        out = mesh.query_points(static_net(), pts, dirs / dirs.norm(dim=-1, keepdim=True), shared_dirs=True)
        assert out.shape == (4 * N, 4) and bool(torch.isfinite(out).all())
        d1 = tight(torch.randn((4 * N, 3), device=dev))
        out = mesh.query_points(static_net(), pts, d1, shared_dirs=False)
    elif case == "sample_coarse":
        rb = ray_batch(11)
        z, pts = render.sample_coarse(rb, 64, True, tight(torch.rand((N, 64), device=dev)), want_pts=True)
        tight(z); tight(pts)
        assert bool(torch.isfinite(pts).all())
if case in ("train_static", "train_dnerf", "train_noview"):
    # one training step at 4096 rays: act / grad [786 432, 2432], xs [.., 96], masks - all tight by their size
    n = 4096
    K, c2w = synth.lego_camera(400, 400)
    o, d = synth.pick_rays(400, 400, K, c2w, n, 2)
    rays = (torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
    tgt = torch.rand((n, 3), device=dev)
    if case == "train_static":
        nets = [static_net(1).train(), static_net(2).train()]
        q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
        rgb, disp, acc, ex = render.render(400, 400, K, rays=rays, ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets[0], network_query_fn=q,
                                           N_samples=64, N_importance=128, network_fine=nets[1], white_bkgd=True, perturb=1., raw_noise_std=1., retraw=True)
        (((rgb - tgt) ** 2).mean() + ((ex["rgb0"] - tgt) ** 2).mean() + 1e-3 * ex["raw"].sum()).backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for m in nets for p in m.parameters())
    elif case == "train_noview":
        nets = [noview_net().train(), noview_net().train()]
        q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=None)
        rgb, disp, acc, ex = render.render(400, 400, K, rays=rays, ndc=False, near=2., far=6., use_viewdirs=False, network_fn=nets[0], network_query_fn=q,
                                           N_samples=64, N_importance=128, network_fine=nets[1], white_bkgd=True, perturb=1., raw_noise_std=1., retraw=True)
        assert ex["raw"].shape == (n, 192, 5)
        (((rgb - tgt) ** 2).mean() + ((ex["rgb0"] - tgt) ** 2).mean() + 1e-3 * ex["raw"].sum()).backward()
        used = [p for m in nets for k, p in m.named_parameters() if k.startswith(("pts_linears", "output_linear"))]
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0 for p in used)
    else:
        dn = dnerf_net().train()
        qd = lambda inputs, viewdirs, ts, network_fn: render_dnerf.run_network(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                                                               embedtime_fn=embedtime_fn, embd_time_discr=True)
        rgb, disp, acc, ex = render_dnerf.render(400, 400, float(K[0][0]), rays=rays, frame_time=0.5, ndc=False, near=2., far=6., use_viewdirs=True,
                                                 network_fn=dn, network_query_fn=qd, N_samples=64, N_importance=128, white_bkgd=True, perturb=1.,
                                                 raw_noise_std=0., retraw=True)
        (((rgb - tgt) ** 2).mean() + 0.1 * ex["position_delta"].pow(2).mean()).backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in dn.parameters())
if case in ("generic_w256_views", "generic_d6"):
    # the layer-by-layer path (swnerf/generic.py): forward + backward on a tight [M, 90] input; every activation and gradient
    # [M, 256 / 283 / 128 ...] is tight by its own size at M = 2^19 (283 x 4 B x 2^19 = 283 x 2 MB)
    M = 1 << 19
    if case == "generic_w256_views":
        net = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[2, 5], use_viewdirs=True).to(dev)
    else:
        net = model.vallina_NeRF(D=6, W=384, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[3], use_viewdirs=True).to(dev)
    assert not net._is_fused_arch()
    x = tight(torch.randn((M, 90), device=dev))
    out = net(x)
    assert out.shape == (M, 4) and bool(torch.isfinite(out).all())
    (out * torch.randn((M, 4), device=dev)).sum().backward()
    gv = net.views_linears[0].weight.grad
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    assert float(gv[:, -27:].abs().max()) > 0, "the view-direction columns of views_linears.0 got no gradient (round 3's bug)"
torch.cuda.synchronize()
print(f"{sys.argv[1]}: ok", flush=True)
