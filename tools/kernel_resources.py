#!/usr/bin/env python3
"""Register / scratch / static-LDS use of every kernel, read from the code-object metadata that hipcc emits (the .amdhsa
block of `hipcc -S --cuda-device-only`), i.e. what the hardware allocates - rocprofv3's VGPR/AGPR/LDS columns are not
reliable for these kernels (they print VGPR 224 / AGPR 0 / LDS 0 for a kernel that holds 444 registers, 192 of them
AGPRs, and ~131 KB of dynamic LDS).  Dynamic LDS is a launch parameter: listed from the launch code by hand below.
usage: kernel_resources.py > profiles/rNN/kernel_resources.md   (no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

DYN_LDS = {  # bytes per workgroup, from the swnerf_* launch code (csrc/*.hip)
    "render_pass_kernel<false,false>": "107,776 (+28,672 with resampling)", "render_pass_kernel<true,false>": "107,776 (+28,672 with resampling)",
    "render_pass_kernel<false,true>": "129,152 (+28,672 with resampling)", "render_pass_kernel<true,true>": "140,544",
    "render_pass_backward_kernel<false>": "142,336", "render_pass_backward_kernel<true>": "145,408",
    "mlp_forward_kernel": "107,776 (ring 8) / 140,544 (ring 16, training unit)", "query_points_kernel": "107,776",
    "gemm_tn_dma_kernel": "131,072 (+16,384 B2 rider / +8,192 A2 rider)",
    "deform_forward_train_kernel": "140,544", "deform_backward_dx_kernel": "140,544", "mlp_backward_dx_kernel": "140,544",
}
print("| kernel | registers per lane (VGPR + AGPR) | of which AGPR | scratch B/lane | static LDS B | dynamic LDS B per workgroup |")
print("|---|---|---|---|---|---|")
with tempfile.TemporaryDirectory() as d:
    for src in ge.SOURCES:
        out = os.path.join(d, src + ".s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S",
                        "--cuda-device-only", "-o", out, os.path.join(ge.CSRC, src)], check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
            g = lambda k: int(re.search(rf"\.amdhsa_{k} (\S+)", m.group(2)).group(1))
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            name = name.replace("void ", "").replace(", ", ",")
            total, acc_off = g("next_free_vgpr"), g("accum_offset")
            dyn = next((v for k, v in DYN_LDS.items() if name.startswith(k)), "0")
            print(f"| `{name}` ({src}) | {total} | {max(0, total - acc_off)} | {g('private_segment_fixed_size')} | {g('group_segment_fixed_size')} | {dyn} |")
