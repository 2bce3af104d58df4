"""Parameter-gradient parity that is immune to ReLU conditioning (test infrastructure; uses the CPU oracle).

A hidden unit whose pre-activation lies within fp32 rounding of zero takes either side of the ReLU kink depending on the
last bits of the evaluation (the CPU's or the kernel's sin / fma order).  One flipped unit moves every gradient below it
by up to ~1e-2 of its max when the loss concentrates on few samples - the fp32 CPU oracle is that far from its own float64
evaluation on such inputs - so a plain tolerance would have to be that loose.  Instead:

  * the truth is the float64 evaluation of the net (on the fp32 encodings the reference forms, nerf/run.py:385,
    embedder.py:33-42) and of raw2outputs,
  * every "risky" unit (|pre-activation| < thr in float64) gets the exact effect of flipping its mask, D_k (rays are
    independent, so it is the difference of two evaluations of its ray),
  * the kernel's gradient must equal truth + sum_k c_k D_k with every c_k in {0, 1}, up to `rtol` of each tensor's max.

That pins the arithmetic at ~1e-5 (fp32 summation noise) instead of 5e-4 .. 1e-2."""
import numpy as np
import torch

from oracle import nerf_oracle as O


def _mlp64(sd, e, flips=None, pres=None):
    """vallina_NeRF.forward (model.py:39-62) in float64 with the ReLU written as a mask; flips: {layer: bool [rows, units]}
    (layers 0..7 = pts_linears, 8 = views_linears.0).  use_viewdirs=False (model.py:59-60) when sd has output_linear and e is
    63 wide; else e = [gamma(x) 63 | gamma(d) 27]."""
    lin = lambda name, x: x @ sd[name + ".weight"].T + sd[name + ".bias"]

    def relu(i, pre):
        if pres is not None:
            pres.append(pre.detach())
        m = pre.detach() > 0
        if flips is not None and i in flips:
            m = m ^ flips[i]
        return pre * m
    pts = e[:, :63]
    h = pts
    for i in range(8):
        h = relu(i, lin(f"pts_linears.{i}", h))
        if i == 4:
            h = torch.cat([pts, h], -1)
    if e.shape[1] == 63:
        return lin("output_linear", h)
    sigma = lin("alpha_linear", h)
    hv = relu(8, lin("views_linears.0", torch.cat([lin("feature_linear", h), e[:, 63:]], -1)))
    return torch.cat([lin("rgb_linear", hv), sigma], -1)


def flip_aware_check(sd_np, rb, z, white_bkgd, ray_loss, gpu_grads, what, thr=5e-6, rtol=2e-5, noise=None):
    """sd_np: the net's fp32 weights (numpy; vallina_NeRF names, with or without view directions); rb [n,8|11], z [n,S] fp32 CPU tensors;
    ray_loss(ret, idx) -> scalar: the loss restricted to rays idx (ret: rgb_map disp_map acc_map raw of those rays) - the
    total loss must be the sum of it over a partition of the rays; gpu_grads: {name: tensor}.  Returns (#flips, #risky)."""
    n, S = z.shape
    names = [k for k in gpu_grads if gpu_grads[k] is not None]
    sd = {k: v.double().requires_grad_(True) for k, v in O.to_torch_sd(sd_np).items()}
    pts = rb[:, None, 0:3] + rb[:, None, 3:6] * z[..., None]
    e_all = O.embed(pts.reshape(-1, 3), 10)
    if rb.shape[1] > 8:                                   # nerf/run.py:80-82: the view directions, expanded per sample
        e_all = torch.cat([e_all, O.embed(rb[:, None, -3:].expand(pts.shape).reshape(-1, 3), 4)], -1)
    e_all = e_all.double().reshape(n, S, -1)

    def grads(idx, flips=None, pres=None):
        for v in sd.values():
            v.grad = None
        raw = _mlp64(sd, e_all[idx].reshape(len(idx) * S, -1), flips, pres).reshape(len(idx), S, -1)
        raw_c = raw[..., :4] if noise is None else torch.cat([raw[..., :3], raw[..., 3:4] + noise[idx].double()[..., None]], -1)
        rgb, disp, acc, _, _ = O.raw2outputs(raw_c, z[idx].double(), rb[idx, 3:6].double(), 0., white_bkgd)
        ray_loss({"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "raw": raw}, idx).backward()
        return torch.cat([(sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])).reshape(-1) for k in names])

    pres = []
    everything = torch.arange(n)
    truth = grads(everything, pres=pres)
    risky = [(l, int(r), int(u)) for l, p in enumerate(pres) for r, u in torch.nonzero(p.abs() < thr).tolist()]
    assert len(risky) <= 400, f"{what}: {len(risky)} units within {thr} of the kink - pick better conditioned inputs"
    cols = []
    for l, row, u in risky:
        ray = torch.tensor([row // S])
        f = torch.zeros((S, pres[l].shape[1]), dtype=torch.bool)
        f[row % S, u] = True
        cols.append(grads(ray, {l: f}) - grads(ray))
    ours = torch.cat([gpu_grads[k].detach().double().cpu().reshape(-1) for k in names])
    diff = ours - truth
    flips = 0
    if cols:
        Dm = torch.stack(cols, 1)
        live = Dm.abs().max(0).values > 1e-3 * rtol * truth.abs().max()       # a flip of a unit no gradient reaches cannot be told
        Dm, units = Dm[:, live], [risky[i] for i in torch.nonzero(live)[:, 0].tolist()]
        if units:
            c = torch.from_numpy(np.linalg.lstsq(Dm.numpy(), diff.numpy()[:, None], rcond=None)[0][:, 0])
            cr = c.round().clamp(0, 1)
            # a coefficient off 0 / 1 only matters if its column is large enough to be told from fp32 summation noise: a unit whose
            # flip moves no tensor by a quarter of the tolerance fits that noise with any coefficient (the residual check below
            # still holds every tensor to rtol with the rounded coefficients)
            amb = (c - cr).abs() * Dm.abs().max(0).values
            bad = ((c - cr).abs() > 0.05) & (amb > 0.25 * rtol * truth.abs().max())
            assert not bool(bad.any()), f"{what}: flip coefficients {c[bad].tolist()} are not 0 / 1 (units {[units[i] for i in torch.nonzero(bad)[:, 0].tolist()]})"
            diff = diff - Dm @ cr
            flips = int(cr.sum())
    o = 0
    for k in names:
        m = gpu_grads[k].numel()
        d, scale = float(diff[o:o + m].abs().max()), max(float(truth[o:o + m].abs().max()), 1e-12)
        assert d <= rtol * scale, f"{what} {k}: {d:.3e} of {scale:.3e} ({d / scale:.2e}) after accounting for {flips} ReLU flips of {len(risky)} risky units"
        o += m
    return flips, len(risky)


noview_flip_aware_check = flip_aware_check
