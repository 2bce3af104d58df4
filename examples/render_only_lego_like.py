#!/usr/bin/env python3
"""The reference's `--render_only` flow (nerf/run.py:545-575) on the MI355X path, end to end, with what exists offline:
a checkpoint in the reference's `.tar` format (synthetic seeded weights - no trained lego checkpoint is available),
`create_nerf` with the option names of configs/lego.txt, the spherical test path of load_blender.py, `render_path`
writing '{:03d}.png' frames.  Everything between the checkpoint and the PNGs runs on the fused HIP pass.

  python examples/render_only_lego_like.py [out_dir] [H=400] [n_poses=4]
  SWNERF_PRECISION=bf16x3-fine python examples/render_only_lego_like.py ...     # the opt-in bf16x3 arithmetic
"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch


def main(out_dir, H=400, n_poses=4, device="cuda:0"):
    from swnerf import synth, runner, render, cameras, checkpoint, model
    dev = torch.device(device)
    W = H
    os.makedirs(out_dir, exist_ok=True)
    # --- a checkpoint as the reference's train() writes it (nerf/run.py:716-724): here from the seeded synthetic nets
    nets = []
    for seed, ab in (synth.NET_COARSE, synth.NET_FINE):
        m = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(seed, alpha_bias=ab).items()})
        nets.append(m)
    checkpoint.save_checkpoint(out_dir, "lego_like", 200000, 200001, nets[0], nets[1], None)
    # --- what nerf/run.py does with configs/lego.txt and --render_only
    args = SimpleNamespace(expname="lego_like", basedir=out_dir, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                           lrate=5e-4, netchunk=1024 * 64, no_reload=False, ft_path=None, N_samples=64, N_importance=128, perturb=1.,
                           use_viewdirs=True, i_embed=0, multires=10, multires_views=4, raw_noise_std=0., dataset_type="blender",
                           white_bkgd=True, no_ndc=False, lindisp=False, chunk=1024 * 32)
    train_kw, test_kw, start, grad_vars, optimizer = runner.create_nerf(args, device=dev)
    assert start == 200001, "the checkpoint was not reloaded"
    test_kw.update(near=2., far=6.)                                            # blender bounds (nerf/run.py:466-467, 525-529)
    H, W, focal = cameras.blender_hwf(H, W, synth.LEGO_CAMERA_ANGLE_X)        # load_blender.py:132-141
    K = cameras.intrinsics(H, W, focal)                                        # nerf/run.py:518-523
    poses = torch.from_numpy(cameras.blender_render_poses(n_poses)).to(dev)    # load_blender.py:136
    frames = os.path.join(out_dir, "renderonly_path_{:06d}".format(start))
    os.makedirs(frames, exist_ok=True)
    with torch.no_grad():
        render.render_path(poses[:1], (H, W, focal), K, args.chunk, test_kw)                       # warm-up (weight packing)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        rgbs, disps = render.render_path(poses, (H, W, focal), K, args.chunk, test_kw, savedir=frames)
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    print(f"rendered {n_poses} frames of {H}x{W} (64+128 samples) in {dt:.2f} s = {n_poses * H * W / dt:,.0f} rays/s incl. PNG writing -> {frames}")
    return rgbs, disps, frames


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/swnerf_example"
    main(out, int(sys.argv[2]) if len(sys.argv) > 2 else 400, int(sys.argv[3]) if len(sys.argv) > 3 else 4)
