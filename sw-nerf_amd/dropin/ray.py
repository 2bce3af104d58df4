"""Shadow of the reference's root-level ray.py: put this directory FIRST on sys.path (before the
reference root) and `from ray import *` in nerf/run.py / d_nerf/run_dnerf.py resolves here.
Re-exports torch/nn/F/np like the reference module does (run_dnerf.py gets them via star-import)."""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch                      # noqa: F401,E402
import torch.nn as nn             # noqa: F401,E402
import torch.nn.functional as F   # noqa: F401,E402
import numpy as np                # noqa: F401,E402
from swnerf.ray import (get_rays, get_rays_np, ndc_rays, sample_pdf, raw2outputs,   # noqa: F401,E402
                        img2mse, mse2psnr, to8b)
