"""examples/ run as written (in process): the reference's --render_only flow end to end on the GPU path - reference-format
checkpoint -> create_nerf reload -> spherical test path -> render_path -> PNG frames."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_render_only_example(tmp_path):
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import render_only_lego_like as ex
    import swnerf.ray as ray
    rgbs, disps, frames = ex.main(str(tmp_path), H=48, n_poses=3)
    assert rgbs.shape == (3, 48, 48, 3) and disps.shape == (3, 48, 48) and np.isfinite(rgbs).all()
    assert 0.0 <= rgbs.min() and rgbs.max() <= 1.0 + 1e-5
    assert float(np.abs(rgbs[0] - rgbs[1]).max()) > 1e-3                      # different poses, different frames
    files = sorted(os.listdir(frames))
    assert files == ["000.png", "001.png", "002.png"]
    assert np.array_equal(np.asarray(Image.open(os.path.join(frames, "001.png"))), ray.to8b(rgbs[1]))
    assert os.path.exists(os.path.join(str(tmp_path), "lego_like", "200000.tar"))
