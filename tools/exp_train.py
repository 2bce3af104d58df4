#!/usr/bin/env python3
"""Timing experiments with an alternative build of the library on the training step: exp_train.py <lib.so> [bench_train.py args]
(tools/experiments/train/build.sh; such builds compute WRONG gradients - timing only)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
from swnerf import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench_train.py"] + sys.argv[2:]
import runpy
runpy.run_path(os.path.join(ROOT, "tools", "bench_train.py"), run_name="__main__")
