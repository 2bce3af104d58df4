"""Shadow of the reference's root-level model.py (see dropin/ray.py).  TNeRF is not provided
(T-NeRF is out of scope)."""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch                      # noqa: F401,E402
import torch.nn as nn             # noqa: F401,E402
import torch.nn.functional as F   # noqa: F401,E402
import numpy as np                # noqa: F401,E402
from swnerf.model import (vallina_NeRF, NeRFOriginal, DirectTemporalNeRF, NeRF,   # noqa: F401,E402
                          img2mse, mse2psnr, to8b)
