#!/usr/bin/env python3
"""Timing experiments with an alternative build of the library: exp_lib.py <lib.so> [bench.py args].
Runs bench.py's forward measurement against that .so (results of such builds are NOT parity-checked)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import __graft_entry__
__graft_entry__.build = lambda: None
from swnerf import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
