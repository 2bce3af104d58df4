"""INTEGRATION.md, checked against the real runners (build container only: skipped where
/root/reference does not exist, e.g. on the GPU box).  Imports nerf/run.py and
d_nerf/run_dnerf.py with sw-nerf_amd/dropin first on sys.path, so their star-imports resolve to this
build, then runs the reference's OWN create_nerf and checks that its network_query_fn lambda is
recognised by the fused dispatch.  Nothing is executed on a device."""
import importlib.util
import os
import sys
import types
from argparse import Namespace

import pytest

REF = "/root/reference"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def _load_runner(rel, name):
    sys.dont_write_bytecode = True
    for mod in ["imageio", "lpips", "skimage", "skimage.metrics", "cv2", "configargparse", "torch.utils.tensorboard"]:
        try:
            importlib.import_module(mod)
        except Exception:
            sys.modules[mod] = types.ModuleType(mod)
    for attr in ("peak_signal_noise_ratio", "structural_similarity"):
        if not hasattr(sys.modules["skimage.metrics"], attr):
            setattr(sys.modules["skimage.metrics"], attr, None)
    if not hasattr(sys.modules["torch.utils.tensorboard"], "SummaryWriter"):
        sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    os.environ.setdefault("MPLBACKEND", "Agg")
    dropin = os.path.join(ROOT, "sw-nerf_amd", "dropin")
    saved = list(sys.path)
    saved_mods = {k: sys.modules.pop(k) for k in ("ray", "embedder", "model", "utils") if k in sys.modules}
    sys.path[:0] = [dropin, os.path.join(REF, os.path.dirname(rel)), REF]
    try:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path[:] = saved
        for k in ("ray", "embedder", "model", "utils"):
            sys.modules.pop(k, None)
        sys.modules.update(saved_mods)
    return mod


def test_static_runner_resolves_to_this_build(tmp_path):
    import swnerf.ray, swnerf.embedder, swnerf.model, swnerf.render
    run = _load_runner("nerf/run.py", "ref_run_dropin")
    assert run.get_rays is swnerf.ray.get_rays and run.raw2outputs is swnerf.ray.raw2outputs
    assert run.sample_pdf is swnerf.ray.sample_pdf and run.ndc_rays is swnerf.ray.ndc_rays
    assert run.get_embedder is swnerf.embedder.get_embedder
    assert run.NeRF is swnerf.model.vallina_NeRF                      # `from model import vallina_NeRF as NeRF`
    args = Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8,
                     netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4,
                     basedir=str(tmp_path), expname="x", ft_path=None, no_reload=True, perturb=1., N_samples=64,
                     white_bkgd=True, raw_noise_std=0., dataset_type="blender", no_ndc=False, lindisp=False)
    os.makedirs(tmp_path / "x")
    kw_train, kw_test, start, grad_vars, opt = run.create_nerf(args)
    assert isinstance(kw_test["network_fn"], swnerf.model.vallina_NeRF) and isinstance(kw_test["network_fine"], swnerf.model.vallina_NeRF)
    assert len(grad_vars) == 48 and start == 0
    # the reference's own lambda is recognised -> render_rays would take the fused pass
    import torch
    with torch.no_grad():        # render_only / test-time rendering (nerf/run.py:566-571)
        assert swnerf.render.fused_plan(kw_test["network_query_fn"], [kw_test["network_fn"], kw_test["network_fine"]]) == (10, 4, 0)
    with torch.enable_grad():    # train(): trainable parameters -> the differentiable op path
        assert swnerf.render.fused_plan(kw_train["network_query_fn"], [kw_train["network_fn"], kw_train["network_fine"]]) is None
    assert kw_test["perturb"] is False and kw_test["raw_noise_std"] == 0.
    # kwargs of the reference's render_rays == ours (level-2 patch is signature-compatible)
    import inspect
    assert list(inspect.signature(run.render_rays).parameters) == list(inspect.signature(swnerf.render.render_rays).parameters)
    assert list(inspect.signature(run.render).parameters) == list(inspect.signature(swnerf.render.render).parameters)
    assert list(inspect.signature(run.run_network).parameters) == list(inspect.signature(swnerf.render.run_network).parameters)


def test_dnerf_runner_resolves_to_this_build(tmp_path):
    import inspect
    import swnerf.model, swnerf.render, swnerf.render_dnerf
    drun = _load_runner("d_nerf/run_dnerf.py", "ref_drun_dropin")
    assert drun.NeRF is swnerf.model.NeRF and drun.torch is not None and drun.np is not None     # via star-imports
    args = Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8,
                     netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4,
                     basedir=str(tmp_path), expname="y", ft_path=None, no_reload=True, perturb=1., N_samples=64,
                     white_bkgd=True, raw_noise_std=0., dataset_type="blender", no_ndc=False, lindisp=False,
                     nerf_type="direct_temporal", not_zero_canonical=False, use_two_models_for_fine=False,
                     do_half_precision=False)
    os.makedirs(tmp_path / "y")
    kw_train, kw_test, start, grad_vars, opt = drun.create_nerf(args)
    net = kw_test["network_fn"]
    assert isinstance(net, swnerf.model.DirectTemporalNeRF) and kw_test["network_fine"] is None
    import torch
    with torch.no_grad():
        assert swnerf.render.fused_plan(kw_test["network_query_fn"], [net, None], need_time=True) == (10, 4, 10)
    assert list(inspect.signature(drun.render_rays).parameters) == list(inspect.signature(swnerf.render_dnerf.render_rays).parameters)
    assert list(inspect.signature(drun.render).parameters) == list(inspect.signature(swnerf.render_dnerf.render).parameters)
    assert list(inspect.signature(drun.run_network).parameters) == list(inspect.signature(swnerf.render_dnerf.run_network).parameters)
