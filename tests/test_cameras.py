"""the camera / pose front-end (host numpy) against vectors captured from the
reference's own loaders run on a synthetic on-disk dataset (tests/golden/make_golden_cameras.py)."""
import json
import os

import numpy as np

import cases
from swnerf import cameras, synth


def test_llff_pipeline_matches_reference(golden):
    ref = golden("g9_cameras")
    pb = cases.g9_poses_bounds()
    assert ref["crc"] == cases.checksum(pb)
    poses, bds, render_poses, i_test = cameras.llff_from_poses_bounds(pb, image_hw=(24, 32), factor=8)
    np.testing.assert_allclose(poses, ref["llff_poses_spiral"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(bds, ref["llff_bds_spiral"], rtol=1e-6)
    np.testing.assert_allclose(render_poses, ref["llff_render_spiral"], rtol=1e-5, atol=1e-6)
    assert i_test == int(ref["llff_itest_spiral"][0]) and render_poses.shape == (120, 3, 5)
    assert poses.dtype == np.float32 and poses[0, 0, 4] == 24 and poses[0, 1, 4] == 32
    assert abs(poses[0, 2, 4] - 3260.5 / 8) < 1e-3                      # focal scaled by the down-sampling factor
    # recentred: the average pose is the identity frame
    avg = cameras.poses_avg(poses)
    np.testing.assert_allclose(avg[:3, :3], np.eye(3), atol=1e-5)
    np.testing.assert_allclose(avg[:3, 3], 0, atol=1e-5)
    assert cameras.llff_near_far(bds) == (0., 1.) and cameras.llff_near_far(bds, no_ndc=True)[0] < bds.min()
    zf = cameras.llff_from_poses_bounds(pb, image_hw=(24, 32), factor=8, path_zflat=True)[2]
    assert zf.shape == (60, 3, 5)        # (the reference itself raises TypeError on this branch with numpy >= 1.18)


def test_blender_front_end_matches_reference(golden, tmp_path):
    ref = golden("g9_cameras")
    frames = cases.g9_blender_frames()
    for s, fr in frames.items():
        json.dump({"camera_angle_x": synth.LEGO_CAMERA_ANGLE_X, "frames": fr}, open(tmp_path / f"transforms_{s}.json", "w"))
    poses, i_split, angle = cameras.blender_meta(str(tmp_path), testskip=2)
    np.testing.assert_array_equal(poses, ref["bl_poses_full"])
    assert [len(s) for s in i_split] == list(ref["bl_split_full"])
    for half, tag in ((False, "full"), (True, "half")):
        hwf = cameras.blender_hwf(16, 16, angle, half_res=half)
        np.testing.assert_allclose(np.array(hwf, np.float64), ref[f"bl_hwf_{tag}"], rtol=1e-12)
    rp = cameras.blender_render_poses()
    assert rp.shape == (360, 4, 4)
    np.testing.assert_allclose(rp, ref["bl_render_full"], atol=1e-6)
    K = cameras.intrinsics(800, 800, cameras.blender_hwf(800, 800, angle)[2])
    assert abs(K[0, 0] - 1111.111) < 1e-2 and K[0, 2] == 400 and K[1, 2] == 400
