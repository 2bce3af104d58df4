"""reference-format checkpoints load into the swnerf modules (and back)."""
import os
import sys

import numpy as np
import pytest
import torch

import cases
from swnerf import checkpoint, model, synth

REF = "/root/reference"


def _nets():
    mk = lambda: model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    return mk(), mk()


def test_roundtrip_and_discovery(tmp_path):
    a, b = _nets()
    opt = torch.optim.Adam(list(a.parameters()) + list(b.parameters()), lr=5e-4, betas=(0.9, 0.999))
    assert checkpoint.reload_latest(str(tmp_path), "exp", a, b, opt) == (0, None)
    p1 = checkpoint.save_checkpoint(str(tmp_path), "exp", 10000, 10001, a, b, opt)
    with torch.no_grad():
        a.rgb_linear.bias.add_(1.0)
    p2 = checkpoint.save_checkpoint(str(tmp_path), "exp", 20000, 20001, a, b, opt)
    assert os.path.basename(p1) == "010000.tar" and checkpoint.find_checkpoints(str(tmp_path), "exp") == [p1, p2]
    assert set(torch.load(p2, weights_only=False).keys()) == {"global_step", "network_fn_state_dict", "network_fine_state_dict", "optimizer_state_dict"}
    c, d = _nets()
    step, path = checkpoint.reload_latest(str(tmp_path), "exp", c, d, None)
    assert (step, path) == (20001, p2)
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), c.state_dict().values()))
    assert checkpoint.reload_latest(str(tmp_path), "exp", c, d, None, no_reload=True) == (0, None)
    assert checkpoint.reload_latest(str(tmp_path), "exp", c, d, None, ft_path=p1)[0] == 10001
    assert checkpoint.to8b(np.array([-1., 0.5, 2.])).tolist() == [0, 127, 255]
    # saved WITHOUT an optimizer: no 'optimizer_state_dict' key, and loading it into a run that has an optimizer keeps that
    # optimizer's fresh state (round-1 advisor finding: an empty dict was written and then fed to load_state_dict)
    p3 = checkpoint.save_checkpoint(str(tmp_path), "exp2", 5, 6, a, b, None)
    assert "optimizer_state_dict" not in torch.load(p3, weights_only=False)
    opt2 = torch.optim.Adam(list(c.parameters()) + list(d.parameters()), lr=1e-3)
    assert checkpoint.load_checkpoint(p3, c, d, opt2) == 6 and opt2.param_groups[0]["lr"] == 1e-3
    torch.save({"global_step": 7, "network_fn_state_dict": a.state_dict(), "network_fine_state_dict": b.state_dict(),
                "optimizer_state_dict": {}}, str(tmp_path / "legacy.tar"))
    assert checkpoint.load_checkpoint(str(tmp_path / "legacy.tar"), c, d, opt2) == 7


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_checkpoint_written_by_the_reference_classes_loads(tmp_path):
    """A .tar written exactly as nerf/run.py:716-724 does, from the reference's own nn.Modules."""
    sys.dont_write_bytecode = True
    saved = sys.modules.pop("model", None)
    sys.path.insert(0, REF)
    try:
        import importlib
        refmodel = importlib.import_module("model")
    finally:
        sys.path.remove(REF)
        sys.modules.pop("model", None)
        if saved is not None:
            sys.modules["model"] = saved
    sd_c, sd_f = cases.weights_static()
    nets = []
    for sd in (sd_c, sd_f):
        m = refmodel.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(m)
    opt = torch.optim.Adam(params=list(nets[0].parameters()) + list(nets[1].parameters()), lr=5e-4, betas=(0.9, 0.999))
    path = tmp_path / "200000.tar"
    torch.save({'global_step': 200000, 'network_fn_state_dict': nets[0].state_dict(),
                'network_fine_state_dict': nets[1].state_dict(), 'optimizer_state_dict': opt.state_dict()}, path)
    a, b = _nets()
    assert checkpoint.load_checkpoint(str(path), a, b) == 200000
    for ours, sd in ((a, sd_c), (b, sd_f)):
        got = ours.state_dict()
        assert list(got.keys()) == list(sd.keys()) and all(np.array_equal(got[k].numpy(), sd[k]) for k in sd)
    dn = refmodel.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27,
                                   input_ch_time=21, use_viewdirs=True, embed_fn=None, zero_canonical=True)
    dn.load_state_dict({k: torch.from_numpy(v) for k, v in cases.weights_dnerf().items()})
    path = tmp_path / "800000.tar"
    torch.save({'global_step': 800000, 'network_fn_state_dict': dn.state_dict(), 'optimizer_state_dict': {}}, path)   # run_dnerf.py:757-770
    ours = model.DirectTemporalNeRF(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21, skips=[4], use_viewdirs=True)
    assert checkpoint.load_checkpoint(str(path), ours) == 800000
    assert all(np.array_equal(ours.state_dict()[k].numpy(), v) for k, v in cases.weights_dnerf().items())


def test_png_writer_roundtrip(tmp_path):
    """The output side of render_path (nerf/run.py:210-213): to8b + PNG, decoded back with PIL."""
    import numpy as np
    from PIL import Image
    from swnerf.png import write_png
    from swnerf.embedder import to8b
    rng = np.random.default_rng(3)
    for shape in ((17, 23, 3), (5, 7, 4), (9, 4), (1, 1, 3), (6, 6, 1)):
        img = to8b(rng.uniform(-0.2, 1.2, shape).astype(np.float32))
        p = str(tmp_path / ("x%d.png" % len(shape) + str(shape[0])))
        write_png(p, img)
        back = np.asarray(Image.open(p))
        assert np.array_equal(back, img[..., 0] if (img.ndim == 3 and img.shape[2] == 1) else img), shape
    import pytest
    with pytest.raises(TypeError):
        write_png(str(tmp_path / "f.png"), np.zeros((2, 2, 3), np.float32))
    with pytest.raises(ValueError):
        write_png(str(tmp_path / "g.png"), np.zeros((2, 2, 5), np.uint8))


def _args(tmp_path, **over):
    """The options create_nerf reads, at the values of the shipped lego / bouncingballs configs
    (nerf/configs/lego.txt, d_nerf/configs/bouncingballs.txt; defaults of utils.py:20-100)."""
    from types import SimpleNamespace
    a = dict(expname="exp", basedir=str(tmp_path), netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, lrate=5e-4,
             netchunk=1024 * 64, no_reload=False, ft_path=None, N_samples=64, N_importance=128, perturb=1., use_viewdirs=True,
             i_embed=0, multires=10, multires_views=4, raw_noise_std=0., dataset_type="blender", white_bkgd=True, no_ndc=False,
             lindisp=False, nerf_type="direct_temporal", not_zero_canonical=False, use_two_models_for_fine=False,
             do_half_precision=False)
    a.update(over)
    return SimpleNamespace(**a)


def test_create_nerf_mirrors_the_runners(tmp_path):
    """nerf/run.py:222-313 and d_nerf/run_dnerf.py:238-352: same return tuple, same render_kwargs keys, the closure is
    recognised by the fused dispatch, checkpoints written in the reference's format are reloaded."""
    import torch
    from swnerf import runner, render, model, checkpoint
    args = _args(tmp_path)
    train, test, start, grad_vars, opt = runner.create_nerf(args, device="cpu")
    assert start == 0 and isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 5e-4 and opt.defaults["betas"] == (0.9, 0.999)
    assert list(train.keys()) == ["network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn",
                                  "use_viewdirs", "white_bkgd", "raw_noise_std", "ndc", "lindisp"]      # run.py:283-299
    assert test["perturb"] is False and test["raw_noise_std"] == 0. and train["perturb"] == 1.
    assert isinstance(train["network_fn"], model.vallina_NeRF) and isinstance(train["network_fine"], model.vallina_NeRF)
    assert train["network_fn"].input_ch == 63 and train["network_fn"].input_ch_views == 27
    assert len(grad_vars) == 48 and sum(p.numel() for p in grad_vars) == 2 * 595844
    with torch.no_grad():
        assert render.fused_plan(train["network_query_fn"], [train["network_fn"], train["network_fine"]]) == (10, 4, 0)
    llff = runner.create_nerf(_args(tmp_path, dataset_type="llff", N_importance=0), device="cpu")
    assert "ndc" not in llff[0] and llff[0]["network_fine"] is None and len(llff[3]) == 24                # run.py:296
    # reload: newest *.tar of basedir/expname (run.py:261-280)
    with torch.no_grad():
        for p in grad_vars:
            p.add_(0.25)
    checkpoint.save_checkpoint(str(tmp_path), "exp", 1000, 1001, train["network_fn"], train["network_fine"], opt)
    t2, _, start2, gv2, _ = runner.create_nerf(args, device="cpu")
    assert start2 == 1001 and all(torch.equal(a, b) for a, b in zip(gv2, grad_vars))
    assert runner.create_nerf(_args(tmp_path, no_reload=True), device="cpu")[2] == 0
    # D-NeRF
    dargs = _args(tmp_path, expname="dn")
    tr, te, st, gv, op = runner.create_dnerf(dargs, device="cpu")
    assert isinstance(tr["network_fn"], model.DirectTemporalNeRF) and tr["network_fine"] is None and st == 0
    assert list(tr.keys())[:10] == ["network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn",
                                    "use_viewdirs", "white_bkgd", "raw_noise_std", "use_two_models_for_fine"]
    assert tr["network_fn"].input_ch_time == 21 and tr["network_fn"].zero_canonical and len(gv) == 42
    with torch.no_grad():
        assert render.fused_plan(tr["network_query_fn"], [tr["network_fn"], None], need_time=True) == (10, 4, 10)
    two = runner.create_dnerf(_args(tmp_path, expname="dn2", use_two_models_for_fine=True, not_zero_canonical=True), device="cpu")
    assert isinstance(two[0]["network_fine"], model.DirectTemporalNeRF) and not two[0]["network_fn"].zero_canonical
    import pytest
    with pytest.raises(NotImplementedError):
        runner.create_dnerf(_args(tmp_path, do_half_precision=True), device="cpu")
