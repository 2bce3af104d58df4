"""Synthetic weights, cameras and ray batches for the parity tests and bench.py.

No dataset or checkpoint exists offline, so every test and
benchmark input is generated here from numpy PCG64 seeds.  Nothing in this file
touches torch RNG, the HIP library or the oracle: it is plain numpy so the same
bytes are produced in the build container and on the GPU box.

Conventions restated from the reference (file:line relative to /root/reference):
  * state_dict key names / shapes of the 8x256 NeRF MLP     model.py:22-37, 251-269
  * DirectTemporalNeRF = `_occ.*` + `_time.{0..7}` + `_time_out`   model.py:108-126
  * pose_spherical(theta, phi, radius)                      dataloader/load_blender.py:11-35
  * K = [[f,0,W/2],[0,f,H/2],[0,0,1]]                       nerf/run.py:518-523
  * blender near/far = 2/6                                  nerf/run.py:466-467
"""
import numpy as np

LEGO_CAMERA_ANGLE_X = 0.6911112070083618   # blender lego transforms_*.json
# canonical synthetic nets: (seed, alpha_linear.bias).  The bias of each net was tuned
# (with the CPU oracle, lego-like rays, near 2 / far 6) so that the accumulated opacity
# has mean ~0.4-0.6 and spans ~(0.05, 1.0); random He-init nets otherwise render fully
# transparent or fully opaque and every parity check on them would be vacuous.
NET_COARSE = (20250310, -0.25)
NET_FINE = (20250311, -1.0)
NET_DNERF = (20250312, -2.0)
D_LAYERS = 8
WIDTH = 256


def _he(rng, out_dim, in_dim):
    return (rng.standard_normal((out_dim, in_dim)) * np.sqrt(2.0 / in_dim)).astype(np.float32)


def nerf_state_dict(seed, input_ch=63, input_ch_views=27, alpha_bias=-0.35, prefix="",
                    bias_scale=0.05):
    """Weights of one vallina_NeRF / NeRFOriginal (use_viewdirs=True) as numpy arrays.

    W ~ N(0, 2/fan_in); small random biases so the bias path is exercised;
    `alpha_linear.bias` is shifted so that the accumulated opacity of a lego-like
    ray spans (0.1, 0.9) instead of saturating."""
    rng = np.random.default_rng(seed)
    W = WIDTH
    sd = {}
    ins = [input_ch] + [W + input_ch if i == 4 else W for i in range(D_LAYERS - 1)]
    for i, k in enumerate(ins):
        sd[f"{prefix}pts_linears.{i}.weight"] = _he(rng, W, k)
        sd[f"{prefix}pts_linears.{i}.bias"] = (rng.standard_normal(W) * bias_scale).astype(np.float32)
    sd[f"{prefix}views_linears.0.weight"] = _he(rng, W // 2, W + input_ch_views)
    sd[f"{prefix}views_linears.0.bias"] = (rng.standard_normal(W // 2) * bias_scale).astype(np.float32)
    sd[f"{prefix}feature_linear.weight"] = _he(rng, W, W)
    sd[f"{prefix}feature_linear.bias"] = (rng.standard_normal(W) * bias_scale).astype(np.float32)
    sd[f"{prefix}alpha_linear.weight"] = _he(rng, 1, W)
    sd[f"{prefix}alpha_linear.bias"] = np.full((1,), alpha_bias, np.float32)
    sd[f"{prefix}rgb_linear.weight"] = _he(rng, 3, W // 2)
    sd[f"{prefix}rgb_linear.bias"] = (rng.standard_normal(3) * bias_scale).astype(np.float32)
    return sd


def noview_state_dict(seed, input_ch=63, output_ch=5, alpha_bias=-0.35, bias_scale=0.05):
    """Weights of one vallina_NeRF with use_viewdirs=False (model.py:22-37, 59-60; the reference's argparse default):
    the 8 trunk layers, the (unused) views_linears.0 that the module still owns, and output_linear [output_ch, 256] whose
    channel 3 (the density) gets the opacity-centring bias."""
    rng = np.random.default_rng(seed)
    W = WIDTH
    sd = {}
    ins = [input_ch] + [W + input_ch if i == 4 else W for i in range(D_LAYERS - 1)]
    for i, k in enumerate(ins):
        sd[f"pts_linears.{i}.weight"] = _he(rng, W, k)
        sd[f"pts_linears.{i}.bias"] = (rng.standard_normal(W) * bias_scale).astype(np.float32)
    sd["views_linears.0.weight"] = _he(rng, W // 2, W)
    sd["views_linears.0.bias"] = (rng.standard_normal(W // 2) * bias_scale).astype(np.float32)
    sd["output_linear.weight"] = _he(rng, output_ch, W)
    sd["output_linear.bias"] = (rng.standard_normal(output_ch) * bias_scale).astype(np.float32)
    sd["output_linear.bias"][3] = alpha_bias
    return sd


def dnerf_state_dict(seed, input_ch=63, input_ch_views=27, input_ch_time=21, alpha_bias=-0.35,
                     dx_scale=0.05, bias_scale=0.05):
    """Weights of one DirectTemporalNeRF: canonical `_occ.*` plus deformation net."""
    sd = nerf_state_dict(seed, input_ch, input_ch_views, alpha_bias, prefix="_occ.",
                         bias_scale=bias_scale)
    rng = np.random.default_rng(seed + 7919)
    W = WIDTH
    ins = [input_ch + input_ch_time] + [W + input_ch if i == 4 else W for i in range(D_LAYERS - 1)]
    for i, k in enumerate(ins):
        sd[f"_time.{i}.weight"] = _he(rng, W, k)
        sd[f"_time.{i}.bias"] = (rng.standard_normal(W) * bias_scale).astype(np.float32)
    # keep |dx| ~ a few cm so x+dx stays inside the scene box
    sd["_time_out.weight"] = (_he(rng, 3, W) * dx_scale).astype(np.float32)
    sd["_time_out.bias"] = (rng.standard_normal(3) * 0.01).astype(np.float32)
    return sd


def pose_spherical(theta_deg, phi_deg, radius):
    """c2w [4,4] float32 on a sphere looking at the origin (load_blender.py:11-35)."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    trans = np.eye(4, dtype=np.float32)
    trans[2, 3] = radius
    rphi = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0],
                     [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], np.float32)
    rth = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0],
                    [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], np.float32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    return (flip @ (rth @ (rphi @ trans))).astype(np.float32)


def lego_camera(H=800, W=800, theta=30.0, phi=-30.0, radius=4.0):
    """(K[3,3] float64, c2w[3,4] float32) of a lego-like blender view."""
    focal = 0.5 * W / np.tan(0.5 * LEGO_CAMERA_ANGLE_X)
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], np.float64)
    return K, pose_spherical(theta, phi, radius)[:3, :4].copy()


def fern_camera(H=378, W=504, focal=407.5658):
    """Forward-facing LLFF-like view."""
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], np.float64)
    c2w = np.eye(4, dtype=np.float32)[:3, :4].copy()
    c2w[0, 3], c2w[1, 3] = 0.05, -0.03
    return K, c2w


def rays_numpy(H, W, K, c2w):
    """All H*W rays of a view, float32 [H*W,3] x2, by the formula of ray.py:42-72
    (used only to fabricate benchmark/test INPUTS; parity of get_rays itself is
    tested separately against the oracle)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., None, :] * c2w[:3, :3], -1).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape).astype(np.float32)
    return rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)


def pick_rays(H, W, K, c2w, n, seed):
    """n distinct rays of the view, chosen by PCG64(seed)."""
    o, d = rays_numpy(H, W, K, c2w)
    sel = np.random.default_rng(seed).choice(H * W, n, replace=False)
    return np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel])


def shard_range(n_total, world_size, rank):
    """Contiguous row-major shard [lo, hi) of a ray index range."""
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
