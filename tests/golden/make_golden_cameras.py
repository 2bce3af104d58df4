#!/usr/bin/env python3
"""Capture g9_cameras.npz: the reference's data loaders (dataloader/load_llff.py, load_blender.py) run,
unmodified, on a tiny SYNTHETIC on-disk dataset written to a temp dir (seeded poses_bounds.npy,
transforms_*.json, dummy image files).  The image reader is stubbed (imageio/cv2 are absent offline)
to return blank images of the right size - nothing on the pose path depends on pixel values."""
import json
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

IMG_HW = {"llff": (24, 32), "blender": (16, 16)}
imageio = types.ModuleType("imageio")
imageio.imread = lambda f: np.zeros(IMG_HW["llff" if "llff" in f else "blender"] + (4,), np.uint8)
sys.modules["imageio"] = imageio
cv2 = types.ModuleType("cv2")
cv2.INTER_AREA = 3
cv2.resize = lambda img, wh, interpolation=None: np.zeros((wh[1], wh[0], img.shape[-1]), img.dtype)
sys.modules["cv2"] = cv2
sys.path.insert(0, "/root/reference")
from dataloader import load_llff, load_blender  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="swnerf_golden_")
    out = {}
    # ---- LLFF
    pb = cases.g9_poses_bounds()
    base = os.path.join(tmp, "llff_scene")
    for d in ("images", "images_8"):
        os.makedirs(os.path.join(base, d))
        for i in range(pb.shape[0]):
            open(os.path.join(base, d, f"img{i:03d}.png"), "wb").close()
    np.save(os.path.join(base, "poses_bounds.npy"), pb)
    for zflat in (False,):      # path_zflat=True raises TypeError in the reference (N_views/=2 -> float into np.linspace)
        images, poses, bds, render_poses, i_test = load_llff.load_llff_data(base, factor=8, recenter=True, bd_factor=.75,
                                                                             spherify=False, path_zflat=zflat)
        tag = "zflat" if zflat else "spiral"
        out.update({f"llff_poses_{tag}": poses, f"llff_bds_{tag}": bds, f"llff_render_{tag}": render_poses,
                    f"llff_itest_{tag}": np.array([i_test])})
    # ---- blender
    base = os.path.join(tmp, "blender_scene")
    frames = cases.g9_blender_frames()
    for s, fr in frames.items():
        os.makedirs(os.path.join(base, s), exist_ok=True)
        for f in fr:
            open(os.path.join(base, f["file_path"] + ".png"), "wb").close()
        json.dump({"camera_angle_x": cases.synth.LEGO_CAMERA_ANGLE_X, "frames": fr}, open(os.path.join(base, f"transforms_{s}.json"), "w"))
    for half in (False, True):
        imgs, poses, render_poses, hwf, i_split = load_blender.load_blender_data(base, half_res=half, testskip=2)
        tag = "half" if half else "full"
        out.update({f"bl_poses_{tag}": poses, f"bl_render_{tag}": render_poses.numpy(), f"bl_hwf_{tag}": np.array(hwf, np.float64),
                    f"bl_split_{tag}": np.concatenate([np.array([len(s)]) for s in i_split])})
    np.savez_compressed(os.path.join(HERE, "g9_cameras.npz"), crc=cases.checksum(pb), **out)
    print("g9_cameras.npz", os.path.getsize(os.path.join(HERE, "g9_cameras.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
