"""GPU gradient parity: hand-written backward kernels against torch autograd
run through the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

import cases
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def relclose(a, b, rtol, atol, what):
    a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    lim = atol + rtol * np.abs(b)
    assert np.all(err <= lim), f"{what}: max err {err.max():.3e} (|ref| up to {np.abs(b).max():.3e}), worst excess {(err-lim).max():.3e}"


@pytest.mark.parametrize("S,white,use_noise", [(64, True, False), (192, False, True), (37, True, True), (2, False, False)])
def test_raw2outputs_backward(dev, S, white, use_noise):
    import swnerf.ray as ray
    rng = np.random.default_rng(500 + S)
    N = 41
    raw = (rng.standard_normal((N, S, 4)) * 1.5).astype(np.float32)
    raw[..., 3] = (rng.standard_normal((N, S)) * 3.0 - 0.5).astype(np.float32)
    raw[0, :, 3] = -2.0                                   # empty ray: every sigma gradient is exactly 0
    z = np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1)
    d = rng.standard_normal((N, 3)).astype(np.float32)
    noise = (rng.standard_normal((N, S)) * 0.5).astype(np.float32) if use_noise else None
    gr, gd, ga, gw, gdep = (rng.standard_normal(s).astype(np.float32) for s in ((N, 3), (N,), (N,), (N, S), (N,)))

    def loss_of(outs, mod):
        rgb, disp, acc, w, depth = outs
        ok = ~torch.isnan(disp)
        return ((rgb * mod(gr)).sum() + (torch.where(ok, disp, torch.zeros_like(disp)) * mod(gd) * 0.05).sum()
                + (acc * mod(ga)).sum() + (w * mod(gw)).sum() + (depth * mod(gdep)).sum())

    r_cpu = T(raw).requires_grad_(True)
    loss_of(O.raw2outputs(r_cpu, T(z), T(d), 0., white, noise=None if noise is None else T(noise)), T).backward()
    r_gpu = T(raw).to(dev).requires_grad_(True)
    outs = ray.raw2outputs(r_gpu, T(z).to(dev), T(d).to(dev), 0, white, noise=None if noise is None else T(noise).to(dev))
    assert outs[0].requires_grad
    loss_of(outs, lambda a: T(a).to(dev)).backward()
    relclose(r_gpu.grad, r_cpu.grad, rtol=2e-4, atol=2e-6, what=f"d raw (S={S})")
    assert float(r_gpu.grad[0, :, 3].abs().max()) == 0.0
    # only d(rgb_map): the loss of nerf/run.py:688-697
    r_cpu.grad = None
    O.raw2outputs(r_cpu, T(z), T(d), 0., white)[0].pow(2).sum().backward()
    r_gpu.grad = None
    ray.raw2outputs(r_gpu, T(z).to(dev), T(d).to(dev), 0, white)[0].pow(2).sum().backward()
    relclose(r_gpu.grad, r_cpu.grad, rtol=2e-4, atol=2e-6, what="d raw from rgb only")
