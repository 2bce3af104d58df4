"""Host-compiled (g++) checks of csrc/swnerf_common.h: the sin/cos used by every kernel, the
linspace restatement, and that the embedding slot maps cover each reference column exactly once."""
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = r'''
#include "swnerf_common.h"
#include <stdio.h>
#include <stdlib.h>
int main() {
  double worst = 0; srand(3);
  for (int it = 0; it < 3000000; ++it) {
    float x = ((float)rand() / RAND_MAX * 2 - 1) * 6.0f; int k = rand() % 10; float y = x * (float)(1 << k);
    for (int c = 0; c < 2; ++c) { double e = fabs((double)sw_sin_or_cos(y, c) - (c ? cos((double)y) : sin((double)y))); if (e > worst) worst = e; }
  }
  printf("%.6e\n", worst);
  int sizes[] = {64, 128, 192, 100, 33, 2};
  for (int s = 0; s < 6; ++s) { for (int i = 0; i < sizes[s]; ++i) printf("%.9g ", sw_linspace(0.f, 1.f, sizes[s], i)); printf("\n"); }
  for (int L = 0; L <= 10; ++L) { for (int a = 0; a < 32; ++a) for (int h = 0; h < 2; ++h) printf("%d ", sw_pos_col(a, h, L)); printf("\n"); }
  for (int L = 0; L <= 4; ++L) { for (int a = 0; a < 16; ++a) for (int h = 0; h < 2; ++h) printf("%d ", sw_dir_col(a, h, L)); printf("\n"); }
  for (int L = 0; L <= 10; ++L) { for (int a = 0; a < 16; ++a) for (int h = 0; h < 2; ++h) printf("%d ", sw_time_col(a, h, L)); printf("\n"); }
  for (int h = 0; h < 2; ++h) { for (int r = 0; r < 16; ++r) printf("%d ", sw_frow(r, h)); printf("\n"); }
  printf("%d %d %d %d\n", SW_CANON_STEPS, SW_DEFORM_STEPS, SW_CANON_FLOATS, SW_DNERF_FLOATS);
  for (int f = 0; f < SW_XS_LD; ++f) printf("%d ", sw_xs_col(f, 10, 4)); printf("\n");
  for (int f = 0; f < SW_XS_LD; ++f) printf("%d ", sw_xs_col(f, 6, 2)); printf("\n");
  printf("%d %d %d\n", SW_BWD_IG_STEPS, SW_BWD_DN_STEPS, SW_XS_LD);
  return 0;
}
'''


def test_common_header_on_host(tmp_path):
    (tmp_path / "t.cpp").write_text(SRC)
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "sw-nerf_amd", "csrc"),
                    str(tmp_path / "t.cpp"), "-o", str(tmp_path / "t"), "-lm"], check=True)
    lines = subprocess.run([str(tmp_path / "t")], check=True, capture_output=True, text=True).stdout.strip().split("\n")
    assert float(lines[0]) < 1.2e-7                                   # max |err| of sin/cos vs double libm, |y| <= 3072
    for ln, n in zip(lines[1:7], (64, 128, 192, 100, 33, 2)):         # == torch.linspace on CPU, bit for bit
        assert np.array_equal(np.array(ln.split(), np.float32), torch.linspace(0., 1., n).numpy())
    k = 7
    for L in range(11):                                               # every column of gamma(x), once
        cols = [int(c) for c in lines[k + L].split() if int(c) >= 0]
        assert sorted(cols) == list(range(3 * (1 + 2 * L)))
    k += 11
    for L in range(5):
        cols = [int(c) for c in lines[k + L].split() if int(c) >= 0]
        assert sorted(cols) == list(range(3 * (1 + 2 * L)))
    k += 5
    for L in range(11):
        cols = [int(c) for c in lines[k + L].split() if int(c) >= 0]
        assert sorted(cols) == list(range(1 + 2 * L))
    k += 11
    rows = [int(c) for c in (lines[k] + " " + lines[k + 1]).split()]
    assert sorted(rows) == list(range(32))                            # the MFMA C/D map is a bijection onto the tile rows
    canon, deform, cf, df = (int(x) for x in lines[k + 2].split())
    # 4 MFMAs per step, 2048 MACs per MFMA on a 32-row tile; the 1-/3-output heads (256+384 resp. 768 MACs per row)
    # run on the VALU, everything else (but the folded feature_linear) is in the stream with <= 1 % padding
    # feature_linear (65536 MACs per row, no activation behind it) is folded into the view layer at pack time: not executed
    assert 0 <= canon * 4 * 2048 - (593408 - 65536 - 256 - 384) * 32 <= 0.01 * 593408 * 32
    assert 0 <= deform * 4 * 2048 - (497152 - 768) * 32 <= 0.01 * 497152 * 32
    assert canon % 8 == 0 and deform % 8 == 0
    # the fused training pass's slot-ordered encodings (xs): slots 0..63 cover gamma(x), 64..95 gamma(d), each reference
    # column exactly once (what swnerf_unslot_grad relies on); the fused D-NeRF backward stream = IG stream + 7 trunk layers
    for ln, (Lp, Ld) in zip(lines[k + 3:k + 5], ((10, 4), (6, 2))):
        cols = [int(c) for c in ln.split()]
        assert len(cols) == 96
        assert sorted(c for c in cols[:64] if c >= 0) == list(range(3 * (1 + 2 * Lp)))
        assert sorted(c for c in cols[64:] if c >= 0) == list(range(3 * (1 + 2 * Ld)))
    ig, dn, xld = (int(x) for x in lines[k + 5].split())
    assert dn == ig + 7 * 256 and dn % 16 == 0 and xld == 96


def test_embed_kernel_multiply_shift_divisions_are_exact():
    """csrc/misc_kernels.hip embed_kernel maps job j of a workgroup to (row, slot) with (j * ceil(2^24 / Q)) >> 24 in 32-bit
    arithmetic, and slot p to (band, component) with (p * ceil(2^24 / d)) >> 24: exact - and free of 32-bit overflow - for every
    (d <= 16, L <= 24) swnerf_embed accepts, at the rows-per-workgroup its launch picks (EMB_ROWS = 64, halved while the image exceeds 64 KB)."""
    for d in range(1, 17):
        d_magic = ((1 << 24) + d - 1) // d
        for L in range(0, 25):
            C, Q = d * (1 + 2 * L), d * (1 + L)
            R = 64
            while R > 1 and R * C * 4 > 64 * 1024:
                R >>= 1
            q_magic = ((1 << 24) + Q - 1) // Q
            j = np.arange(R * Q, dtype=np.uint64)
            assert int(j[-1]) * q_magic < 2 ** 32, (d, L, R)
            assert np.array_equal((j * np.uint64(q_magic)) >> np.uint64(24), j // np.uint64(Q)), (d, L, R)
            pslot = np.arange(max(d * L, 1), dtype=np.uint64)
            assert int(pslot[-1]) * d_magic < 2 ** 32
            assert np.array_equal((pslot * np.uint64(d_magic)) >> np.uint64(24), pslot // np.uint64(d)), (d, L)
