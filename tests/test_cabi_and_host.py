"""CPU-only checks: the C-ABI library loads and exports every symbol include/swnerf.h declares,
the ctypes mirror of swnerf_pass_args has the C layout, and the host-side logic (shards,
fused-dispatch detection, argument validation, loud failure without a GPU) behaves."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "swnerf.h")


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.build()
    from swnerf import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    text = open(HEADER).read()
    declared = sorted(set(re.findall(r"\b(swnerf_[a-z_0-9]+)\s*\(", text)))
    assert len(declared) >= 20
    L = ctypes.CDLL(built.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in swnerf.h but not exported"
    assert sorted(built.EXPORTS) == declared
    assert L.swnerf_version() == int(re.search(r"#define SWNERF_VERSION (\d+)", text).group(1))


def test_pass_args_layout_matches_c(built, tmp_path):
    """sizeof / offsetof of swnerf_pass_args as gcc sees them == the ctypes Structure."""
    fields = [f for f, _ in built.PassArgs._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "swnerf.h"\nint main(){\n'
                   'printf("%zu\\n", sizeof(swnerf_pass_args));\n'
                   + "".join(f'printf("%zu\\n", offsetof(swnerf_pass_args, {f}));\n' for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    nums = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(built.PassArgs)
    assert nums[1:] == [getattr(built.PassArgs, f).offset for f in fields]


def test_packed_sizes_and_argument_errors_without_gpu(built):
    L = built.lib()
    fold = 128 * 288 + 128                                             # W_vf | Wv[:, 256:] and b_vf: feature_linear folded into the view layer
    canon = (2064 + 16) * 256 + 89 * 32 + (144 + 16) * 256 + fold      # stream + tail, bias/head tiles, views loop, fold
    dnerf = (1952 + 2064 + 16) * 256 + (89 + 89) * 32 + canon
    assert L.swnerf_packed_floats(0) == canon and L.swnerf_packed_floats(1) == dnerf and L.swnerf_packed_floats(7) == 0
    # pure argument validation happens before any device call
    assert L.swnerf_render_pass(None, None) == -1 and b"NULL" in L.swnerf_last_error()
    assert L.swnerf_embed(None, 4, 3, 10, None, None) == -1
    assert L.swnerf_raw2outputs(None, None, None, None, 4, 1, 0, None, None, None, None, None, None) == -2
    assert b"degenerate" in L.swnerf_last_error()
    assert L.swnerf_sample_pdf(None, None, 4, 5000, 8, None, None, None, 0, None, None, None) != 0
    assert L.swnerf_pack_net(0, None, 10, 4, 0, None, None) == -1
    # training entry points: stream sizes per kind, bit-mask buffer size, NULL / bad-kind rejection
    bwd = (1936 + 16) * 256 + 8 * 32 + fold                            # RGB^T 16 | W_vf^T 128 | L7^T..L1^T 7 x 256
    assert [L.swnerf_packed_bwd_floats_kind(k) for k in (0, 1, 2, 3, 4)] == [
        bwd, bwd + 128 * 256, (1792 + 16) * 256 + 24 * 32, (1936 + 128 + 1792 + 16) * 256 + 32 * 32 + fold, 0]     # 3: the fused D-NeRF stream
    assert L.swnerf_packed_bwd_floats() == bwd and L.swnerf_act_floats_per_row() == 2432
    assert [L.swnerf_mask_floats(m) for m in (0, 1, 32, 33, 786432)] == [0, 2304, 2304, 4608, 786432 // 32 * 2304]
    assert L.swnerf_pack_net_bwd_kind(5, None, 10, 4, None, None) == -1
    assert L.swnerf_mlp_forward_train(None, None, 4, 10, 4, None, None, None, None) == -1
    assert L.swnerf_mlp_backward_dx(None, None, None, 4, None, None) == -1
    assert L.swnerf_mlp_backward_dx_pts(None, None, None, None, 4, 10, None, None, None) == -1
    assert L.swnerf_deform_forward_train(None, None, None, 4, 10, 4, 10, None, None, None, None) == -1
    assert L.swnerf_deform_backward_dx(None, None, None, 4, None, None) == -1 and b"NULL" in L.swnerf_last_error()
    assert L.swnerf_gemm_tn(None, 4, 1, None, 4, 1, 8, None, 4, None, None) == -1
    # round 2: the fused training passes, the generic layers
    assert [L.swnerf_train_rows(n, s_) for n, s_ in ((4096, 192), (3, 33), (5, 64), (0, 64))] == [786432, 192, 320, 0]
    assert L.swnerf_xs_floats_per_row() == 96
    assert L.swnerf_render_pass_train(None, None, None, None, None) == -1 and b"NULL" in L.swnerf_last_error()
    a = built.PassArgs()
    a.packed, a.ray_batch, a.n_rays, a.cols, a.kind, a.n_samples = 8, 8, 4, 11, 1, 64             # (never dereferenced: rejected first)
    assert L.swnerf_render_pass_train(a, 8, 8, 8, None) == -2 and b"static net" in L.swnerf_last_error()
    a.kind, a.n_samples = 0, 300
    assert L.swnerf_render_pass_train(a, 8, 8, 8, None) == -2 and b"n_samples" in L.swnerf_last_error()
    assert L.swnerf_render_pass_train_dnerf(a, 8, 8, 8, 8, 8, 8, None) == -2 and b"DirectTemporalNeRF" in L.swnerf_last_error()
    assert L.swnerf_render_pass_backward(None, None, None, None, None, 11, None, 4, 64, 0, None, None, None, None, None, None, None) == -1
    assert L.swnerf_render_pass_backward(8, 8, 8, 8, 8, 11, None, 4, 300, 0, None, None, None, None, 8, 8, None) == -2
    assert L.swnerf_render_pass_backward_dnerf(None, None, None, None, None, None, 12, None, None, None, 4, 64, 0, 10, None, None, None, None,
                                               None, None, None, None, None) == -1
    assert L.swnerf_unslot_grad(None, 64, 256, 0, 64, 10, 4, None, 63, 0, None) == -1
    assert L.swnerf_unslot_grad(8, 64, 256, 64, 64, 10, 4, 8, 63, 0, None) == -1                  # slots 64..127 do not exist
    assert L.swnerf_unslot_grad_time(8, 32, 256, 33, 10, 8, 84, 63, None) == -1
    assert L.swnerf_linear(None, 8, 4, 8, None, None, 4, 0, None, 4, None) == -1 and b"linear" in L.swnerf_last_error()
    assert L.swnerf_linear(None, 8, 0, 8, None, None, 4, 0, None, 4, None) == 0                    # M = 0: nothing to do
    assert L.swnerf_gemm_nn(8, 2, 4, 8, 8, 4, 4, 8, 4, None) == -1                                # lda < K
    assert L.swnerf_relu_mask(None, None, 5, None) == -1 and L.swnerf_relu_mask(None, None, 0, None) == 0


def test_no_cpu_fallback(built):
    from swnerf import ray, model, embedder
    with pytest.raises(RuntimeError, match="GPU"):
        ray.raw2outputs(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="GPU"):
        embedder.get_embedder(10, 3)[0](torch.zeros(4, 3))
    m = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(4, 90))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no GPU|No HIP|GPU"):
            ray.get_rays(4, 4, 10.0, torch.eye(4)[:3])


def test_missing_library_is_loud(built, monkeypatch):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libswnerf_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        built.lib()


def test_module_parameter_names_and_shapes(built):
    from swnerf import model, synth
    m = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    ref = synth.nerf_state_dict(1)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: v.shape for k, v in ref.items()}
    assert sum(p.numel() for p in m.parameters()) == 595844
    d = model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27,
                               input_ch_time=21, use_viewdirs=True, embed_fn=None, zero_canonical=True)
    refd = synth.dnerf_state_dict(1)
    assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == {k: v.shape for k, v in refd.items()}
    assert sum(p.numel() for p in d.parameters()) == 1095047
    with pytest.raises(ValueError):
        model.NeRF.get_by_name("nope")
    o = model.NeRFOriginal(D=8, W=256, input_ch=63, input_ch_views=27, use_viewdirs=True)
    assert abs(float(o.pts_linears[1].weight.std()) - np.sqrt(2 / 256)) < 0.01     # kaiming_normal (model.py:270-272)


def test_fused_dispatch_detection(built):
    with torch.no_grad():
        _fused_dispatch_detection()
    # with autograd on and trainable parameters render_rays must take the differentiable op path
    from swnerf import embedder, model, render
    e10, c10 = embedder.get_embedder(10, 3, 0)
    e4, c4 = embedder.get_embedder(4, 3, 0)
    net = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[4], use_viewdirs=True)
    tagged = lambda a, b, c: None
    tagged.swnerf_embedders = {"embed_fn": e10, "embeddirs_fn": e4}
    with torch.enable_grad():
        assert render.fused_plan(tagged, [net]) is None
        for p in net.parameters():
            p.requires_grad_(False)
        assert render.fused_plan(tagged, [net]) == (10, 4, 0)


_g_e10 = _g_e4 = None
# a query lambda written at MODULE level (scripts, notebooks): its encoders are globals, not closure cells
_g_query = lambda inputs, viewdirs, network_fn: None if False else (embed_fn, embeddirs_fn)     # noqa: E731,F821


def test_fused_dispatch_sees_module_level_lambda(built):
    from swnerf import embedder, model, render
    e10, c10 = embedder.get_embedder(10, 3, 0)
    e4, c4 = embedder.get_embedder(4, 3, 0)
    net = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[4], use_viewdirs=True)
    g = globals()
    g["embed_fn"], g["embeddirs_fn"] = e10, e4
    try:
        with torch.no_grad():
            assert render.closure_embedders(_g_query) == {"embed_fn": e10, "embeddirs_fn": e4}
            assert render.fused_plan(_g_query, [net]) == (10, 4, 0)
    finally:
        del g["embed_fn"], g["embeddirs_fn"]


def _fused_dispatch_detection():
    from swnerf import embedder, model, render, render_dnerf
    e10, c10 = embedder.get_embedder(10, 3, 0)
    e4, c4 = embedder.get_embedder(4, 3, 0)
    et, ct = embedder.get_embedder(10, 1, 0)
    assert (c10, c4, ct) == (63, 27, 21) and (e10.multires, e4.multires) == (10, 4)
    net = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[4], use_viewdirs=True)
    embed_fn, embeddirs_fn = e10, e4
    q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                embeddirs_fn=embeddirs_fn, netchunk=65536)
    assert render.fused_plan(q, [net, None]) == (10, 4, 0)
    assert render.fused_plan(lambda a, b, c: None, [net]) is None                 # no encoders in the closure
    embed_fn = lambda x: x                                                          # a foreign encoder
    q2 = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                                 embeddirs_fn=embeddirs_fn, netchunk=65536)
    assert render.fused_plan(q2, [net]) is None
    assert render.fused_plan(q, [torch.nn.Linear(3, 3)]) is None                    # a foreign network
    small = model.vallina_NeRF(D=8, W=256, input_ch=39, input_ch_views=c4, skips=[4], use_viewdirs=True)
    assert render.fused_plan(q, [small]) is None                                    # encoder / net size mismatch
    tagged = lambda a, b, c: None
    tagged.swnerf_embedders = {"embed_fn": e10, "embeddirs_fn": e4}
    assert render.fused_plan(tagged, [net]) == (10, 4, 0)
    dn = model.DirectTemporalNeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, input_ch_time=ct, skips=[4],
                                  use_viewdirs=True, embed_fn=e10)
    embed_fn, embedtime_fn = e10, et
    qd = lambda inputs, viewdirs, ts, network_fn: render_dnerf.run_network(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn,
                                                                           embeddirs_fn=embeddirs_fn, embedtime_fn=embedtime_fn)
    assert render.fused_plan(qd, [dn, None], need_time=True) == (10, 4, 10)
    assert render.fused_plan(q, [dn], need_time=True) is None                       # no time encoder
    with pytest.raises(NotImplementedError):
        embedder.Embedder(include_input=True, input_dims=3, max_freq_log2=9, num_freqs=10, log_sampling=False,
                          periodic_fns=[torch.sin, torch.cos])
    ident, d = embedder.get_embedder(10, 3, -1)
    assert d == 3 and ident(torch.ones(2, 3)).shape == (2, 3)


def test_synth_shards_and_cameras(built):
    from swnerf import synth
    for n, w in ((640000, 8), (160000, 8), (10, 3), (7, 8), (0, 4)):
        rs = [synth.shard_range(n, w, r) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
    assert synth.shard_range(640000, 8, 3) == (240000, 320000)                      # 100 image rows per GPU
    K, c2w = synth.lego_camera(800, 800)
    assert abs(K[0, 0] - 1111.111) < 1e-2 and c2w.shape == (3, 4)
    assert abs(np.linalg.norm(c2w[:, 3]) - 4.0) < 1e-5 and abs(np.linalg.det(c2w[:, :3]) - 1.0) < 1e-5
    o, d = synth.pick_rays(400, 400, *synth.lego_camera(400, 400), 16, seed=1)
    assert o.shape == d.shape == (16, 3) and o.dtype == np.float32
    a, b = synth.nerf_state_dict(5), synth.nerf_state_dict(5)
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_get_rays_np_is_the_oracles(built):
    from swnerf import ray, synth
    from oracle import nerf_oracle as O
    K, c2w = synth.lego_camera(20, 30)
    for f in (K, float(K[0, 0])):
        a, b = ray.get_rays_np(20, 30, f, c2w), O.get_rays_np(20, 30, f, c2w)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_precision_switch_is_validated():
    """render.set_precision / SWNERF_PRECISION accept exactly the documented names (fp32 is the default and the parity path)."""
    import subprocess
    import sys
    import swnerf.render as render
    assert render.PRECISION == "fp32"
    with pytest.raises(ValueError):
        render.set_precision("fp16")
    assert render.set_precision("bf16x3-fine") == "fp32" and render.set_precision("fp32") == "bf16x3-fine"
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, 'sw-nerf_amd'); import swnerf.render"],
                       env=dict(os.environ, SWNERF_PRECISION="tf32"), cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "SWNERF_PRECISION" in r.stderr
