"""Out-of-bounds guard for every C-ABI entry point, as tests (round 3 kept this as a tool and its cases missed the path that
had faulted): tools/tight_buffer_check.py runs each entry point on operands whose allocation ENDS with their last element
(>= 10 MB, a multiple of 2 MB: the caching allocator then maps exactly that much), so a kernel that reads or writes past its
last row leaves the mapping and the process dies with "Memory access fault" instead of silently touching a neighbour - GPU
AddressSanitizer is not available on this pool.  Includes the generic path's shapes: W = 256 with view directions and skips
[2, 5] forward + backward (the 128 x 283 weight gradient that faulted in round 3) and a D != 8 net.

Fresh child processes, started before this pytest process has initialised the GPU (this module sorts FIRST on purpose - in
front of test_00_bench_launcher.py, whose last test initialises the GPU in-process: a process that has may not start programs
on this pool).  A few cases per child: a fault costs that child's remaining cases,
never the suite's process."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TOOL = os.path.join(ROOT, "tools", "tight_buffer_check.py")
GROUPS = [
    ["rays", "embed", "raw2outputs", "sample_pdf", "sample_coarse", "query"],
    ["pass_static", "pass_dnerf", "pass_noview", "mlp_static", "mlp_dnerf", "mlp_noview"],
    ["train_static", "train_noview", "train_dnerf"],
    ["generic_w256_views", "generic_d6", "pass_x3_static", "pass_x3_dnerf"],
]


def test_every_case_is_in_a_group():
    cases = subprocess.run([sys.executable, TOOL, "list"], capture_output=True, text=True, timeout=60).stdout.split()
    assert sorted(cases) == sorted(c for g in GROUPS for c in g)


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("group", GROUPS, ids=lambda g: g[0])
def test_entry_points_on_tight_allocations(group):
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: starting programs from it is not allowed on this pool")
    r = subprocess.run([sys.executable, TOOL] + group, capture_output=True, text=True, timeout=800)
    out = r.stdout + r.stderr
    assert "Memory access fault" not in out and "HSA_STATUS_ERROR" not in out, out[-3000:]
    assert r.returncode == 0, out[-3000:]
    for c in group:
        assert f"{c}: ok" in r.stdout, (c, out[-2000:])
