#!/usr/bin/env python3
"""Training step of a use_viewdirs=False run (the reference's argparse default) at the C2 shape: 4096 rays x (64 + 128), two
8x256 nets, forward + backward (no optimizer), on the fused pass and - SWNERF_TRAIN_OP_PATH=1 - on the layer-by-layer generic
path it replaces.  Prints ms per step and the fraction of the fp32 MFMA peak (3 x 2 x MACs x rows)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import model, render, embedder, synth

dev = torch.device("cuda:0")
N, S, NI = 4096, 64, 128
embed_fn, _ = embedder.get_embedder(10, 3, 0)
q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=None, netchunk=1024 * 64)
kw = dict(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
nets = []
for seed, ab in ((20250321, 0.5), (20250322, 0.7)):
    m = model.vallina_NeRF(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.noview_state_dict(seed, alpha_bias=ab).items()}, strict=True)
    nets.append(m.to(dev).train())
g = torch.Generator().manual_seed(3)
o = torch.randn((N, 3), generator=g) * 0.1 + torch.tensor([0., 0., 4.])
d = torch.nn.functional.normalize(-o + torch.randn((N, 3), generator=g) * 0.5, dim=-1)
rb = torch.cat([o, d, torch.full((N, 1), 2.), torch.full((N, 1), 6.)], -1).to(dev)
tgt = torch.rand((N, 3), generator=g).to(dev)
macs = 63 * 256 + 6 * 256 * 256 + (256 + 63) * 256 + 256 * 5
flops = 3 * 2 * macs * N * (S + S + NI)
print("| path | ms per step (forward + backward) | TFLOP/s | of the 157.3 TFLOP/s fp32 MFMA peak |")
print("|---|---|---|---|")
ONLY_FUSED = len(sys.argv) > 1 and sys.argv[1] == "fused"          # for rocprofv3: 2 warm-up + 7 timed steps of the fused path
for name, env in (("fused (render_pass_train / render_pass_backward_noview)", None), ("generic layer by layer", "1")):
    if ONLY_FUSED and env:
        continue
    if env:
        os.environ["SWNERF_TRAIN_OP_PATH"] = env
    else:
        os.environ.pop("SWNERF_TRAIN_OP_PATH", None)

    def step():
        for n_ in nets:
            for p_ in n_.parameters():
                p_.grad = None
        r = render.render_rays(rb, nets[0], q, S, N_importance=NI, network_fine=nets[1], white_bkgd=True, retraw=True)
        (torch.mean((r["rgb_map"] - tgt) ** 2) + torch.mean((r["rgb0"] - tgt) ** 2)).backward()
    t0 = time.perf_counter()
    w = 0
    while (w < 2) if ONLY_FUSED else (time.perf_counter() - t0 < 0.5):
        step()
        w += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    while (k < 7) if ONLY_FUSED else (k < 5 or time.perf_counter() - t0 < 1.0):
        step()
        k += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print(f"| {name} | {dt * 1e3:.2f} | {flops / dt / 1e12:.1f} | {flops / dt / 157.3e12 * 100:.1f} % |")
print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
