"""Seeded INPUTS of the golden cases G1-G8.

Imported by tests/golden/make_golden.py (which feeds them to the unmodified
reference in the build container and stores the reference's OUTPUTS in *.npz) and by
the tests (which feed the same inputs to the oracle / the HIP path).  numpy PCG64
only, so the GPU box regenerates identical bytes; each .npz also stores a checksum
of the inputs it was made from and the tests verify it.
"""
import os
import sys
import zlib
import numpy as np

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
_PKG = os.path.join(_ROOT, "sw-nerf_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
from swnerf import synth  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))



def legacy_rand(*shape):
    """The draw the reference's own `pytest=True` hook makes (ray.py:124-132,181-184,
    nerf/run.py:378-381): np.random.seed(0); np.random.rand(*shape), then float32."""
    np.random.seed(0)
    return np.random.rand(*shape).astype(np.float32)


def checksum(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.array([c], dtype=np.int64)


def weights_static():
    return (synth.nerf_state_dict(synth.NET_COARSE[0], alpha_bias=synth.NET_COARSE[1]),
            synth.nerf_state_dict(synth.NET_FINE[0], alpha_bias=synth.NET_FINE[1]))


def weights_dnerf():
    return synth.dnerf_state_dict(synth.NET_DNERF[0], alpha_bias=synth.NET_DNERF[1])


# ------------------------------------------------------------------ G1 embed
def g1_inputs():
    rng = np.random.default_rng(101)
    pts = rng.uniform(-6, 6, (1024, 3)).astype(np.float32)
    dirs = rng.standard_normal((1024, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    t = rng.uniform(0, 1, (1024, 1)).astype(np.float32)
    t[0, 0], t[1, 0] = 0.0, 1.0
    return dict(pts=pts, dirs=dirs.astype(np.float32), t=t)


# ------------------------------------------------------------------ G2 rays
def g2_inputs():
    K400, c2w400 = synth.lego_camera(400, 400, theta=30.0)
    K_small, c2w_small = synth.lego_camera(32, 48, theta=-75.0, phi=-20.0)
    Kf, c2wf = synth.fern_camera()
    return dict(K400=K400, c2w400=c2w400, K_small=K_small, c2w_small=c2w_small,
                focal400=float(K400[0, 0]), Kf=Kf, c2wf=c2wf)


# ------------------------------------------------------------------ G3 coarse z
def g3_inputs():
    rng = np.random.default_rng(103)
    n = 256
    K, c2w = synth.lego_camera(400, 400, theta=60.0)
    o, d = synth.pick_rays(400, 400, K, c2w, n, seed=1031)
    near = np.full((n, 1), 2.0, np.float32)
    far = np.full((n, 1), 6.0, np.float32)
    t_rand = legacy_rand(n, 64)
    return dict(rays_o=o, rays_d=d, near=near, far=far, t_rand=t_rand)


# ------------------------------------------------------------------ G4 MLP
def g4_inputs():
    rng = np.random.default_rng(104)
    m = 4096
    x = rng.uniform(-1, 1, (m, 90)).astype(np.float32)
    x[:, :3] = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
    # rows of an exactly embedded point, so the values look like the real thing
    for r in range(0, 64):
        p = x[r, :3]
        e = [p]
        for k in range(10):
            e.append(np.sin(p * np.float32(2.0 ** k)))
            e.append(np.cos(p * np.float32(2.0 ** k)))
        x[r, :63] = np.concatenate(e).astype(np.float32)
    return dict(x=x)


def time_embed_np(t, L=10):
    """[t, sin(2^k t), cos(2^k t)...] for a scalar t as float32 (embedder.py:33-42)."""
    t = np.float32(t)
    e = [np.array([t], np.float32)]
    for k in range(L):
        a = np.float32(t * np.float32(2.0 ** k))
        e.append(np.array([np.sin(a)], np.float32))
        e.append(np.array([np.cos(a)], np.float32))
    return np.concatenate(e).astype(np.float32)


# ------------------------------------------------------------------ G5 raw2outputs
def g5_inputs(S):
    rng = np.random.default_rng(105 + S)
    n = 128
    raw = (rng.standard_normal((n, S, 4)) * 1.5).astype(np.float32)
    raw[..., 3] = (rng.standard_normal((n, S)) * 4.0 - 1.0).astype(np.float32)
    raw[0, :, 3] = -5.0            # empty ray: all-zero density -> acc 0, disp NaN
    raw[1, :, 3] = 200.0           # saturated ray
    raw[2, :, 3] = 0.0
    raw[2, S // 2, 3] = 50.0       # single spike
    z = np.sort(rng.uniform(2, 6, (n, S)).astype(np.float32), axis=-1)
    z[3] = np.linspace(2, 6, S, dtype=np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    noise = legacy_rand(n, S)      # raw_noise_std = 1.0 through the pytest hook
    return dict(raw=raw, z=z, rays_d=d, noise=noise)


# ------------------------------------------------------------------ G6 sample_pdf
def g6_inputs():
    rng = np.random.default_rng(106)
    n, nb = 128, 63
    z = np.sort(rng.uniform(2, 6, (n, 64)).astype(np.float32), axis=-1)
    z[0] = np.linspace(2, 6, 64, dtype=np.float32)
    z[1] = z[0]
    z[2] = z[0]
    bins = (0.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    w = (rng.uniform(0, 1, (n, nb - 1)) ** 4).astype(np.float32)
    w[0] = 0.25                    # all-equal weights
    w[1] = 0.0
    w[1, 30] = 1.0                 # single spike
    w[2] = 0.0                     # all zero (only the 1e-5 floor)
    u = legacy_rand(n, 128)
    return dict(z=z, bins=bins, weights=w, u=u)


# ------------------------------------------------------------------ G7/G8 render_rays
def g7_inputs(n=1024, H=400, W=400, seed=1):
    K, c2w = synth.lego_camera(H, W, theta=30.0)
    o, d = synth.pick_rays(H, W, K, c2w, n, seed)
    return dict(rays_o=o, rays_d=d, near=2.0, far=6.0)


def g7_ndc_inputs(n=256, seed=3):
    K, c2w = synth.fern_camera()
    o, d = synth.pick_rays(378, 504, K, c2w, n, seed)
    return dict(rays_o=o, rays_d=d, near=0.0, far=1.0, H=378, W=504, focal=float(K[0, 0]))


def g7_rand_inputs(n=256):
    return dict(t_rand=legacy_rand(n, 64), u=legacy_rand(n, 128))


def g8_inputs(n=512):
    return g7_inputs(n=n, seed=5)


# ------------------------------------------------------------------ G9 cameras
def g9_poses_bounds(n=11):
    """A seeded LLFF `poses_bounds.npy` ([n,17]): forward-facing cameras jittered around the origin."""
    rng = np.random.default_rng(109)
    rows = []
    for i in range(n):
        ang = rng.normal(0, 0.08, 3)
        cx, sx, cy, sy = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1])
        R = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        t = rng.normal(0, 0.6, 3) * np.array([1.0, 0.6, 0.15])
        hwf = np.array([3024., 4032., 3260.5])
        m = np.concatenate([R, t[:, None], hwf[:, None]], 1)          # [3,5]
        rows.append(np.concatenate([m.reshape(-1), [rng.uniform(3.5, 5.0), rng.uniform(30., 80.)]]))
    return np.array(rows, np.float64)


def g9_blender_frames():
    out = {}
    for s, thetas in (("train", range(0, 360, 40)), ("val", range(10, 360, 90)), ("test", range(5, 360, 60))):
        out[s] = [{"file_path": f"./{s}/r_{i}", "transform_matrix": synth.pose_spherical(float(t), -30.0 + i, 4.0).astype(float).tolist()}
                  for i, t in enumerate(thetas)]
    return out


# ------------------------------------------------------------------ G10 mesh grid query
G10_BOUNDS = [(-1., 1.), (-1., 2.), (-4., 2.)]          # nerf/extract_mesh.py:148
G10_RES, G10_VIEWS = 6, 8


# ------------------------------------------------------------------ G11 any-shape nets (model.py:59-60, utils.py:26-29)
def generic_state_dict(seed, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, prefix="", alpha_bias=0.0):
    """Seeded weights with the reference's parameter names / shapes for ANY (D, W, skips, use_viewdirs) of
    vallina_NeRF / NeRFOriginal (model.py:22-37, 251-269): He-normal weights, small biases."""
    rng = np.random.default_rng(seed)
    he = lambda o, i: (rng.standard_normal((o, i)) * np.sqrt(2.0 / max(i, 1))).astype(np.float32)
    bias = lambda o: (rng.standard_normal(o) * 0.05).astype(np.float32)
    sd = {}
    ins = [input_ch] + [W + input_ch if i in skips else W for i in range(D - 1)]
    for i, k in enumerate(ins):
        sd[f"{prefix}pts_linears.{i}.weight"], sd[f"{prefix}pts_linears.{i}.bias"] = he(W, k), bias(W)
    sd[f"{prefix}views_linears.0.weight"], sd[f"{prefix}views_linears.0.bias"] = he(W // 2, W + input_ch_views), bias(W // 2)
    if use_viewdirs:
        sd[f"{prefix}feature_linear.weight"], sd[f"{prefix}feature_linear.bias"] = he(W, W), bias(W)
        sd[f"{prefix}alpha_linear.weight"], sd[f"{prefix}alpha_linear.bias"] = he(1, W), np.full((1,), alpha_bias, np.float32)
        sd[f"{prefix}rgb_linear.weight"], sd[f"{prefix}rgb_linear.bias"] = he(3, W // 2), bias(3)
    else:
        sd[f"{prefix}output_linear.weight"], sd[f"{prefix}output_linear.bias"] = he(output_ch, W), bias(output_ch)
        sd[f"{prefix}output_linear.bias"][3] = alpha_bias
    return sd


# (name, kwargs): the shapes G11 covers.  novd = what `--use_viewdirs` unset gives (nerf/run.py:226-231: input_ch_views
# = 0, output_ch = 5 with N_importance > 0); small = another depth / width / skip set; the D-NeRF pair at D=4, W=64.
G11_NETS = {
    "novd": dict(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False),
    "small": dict(D=4, W=128, input_ch=63, input_ch_views=27, output_ch=5, skips=[2], use_viewdirs=True),
    "tiny_novd": dict(D=3, W=96, input_ch=39, input_ch_views=0, output_ch=4, skips=[], use_viewdirs=False),
}
G11_DNERF = dict(D=4, W=64, input_ch=63, input_ch_views=27, input_ch_time=21, output_ch=5, skips=[1], use_viewdirs=True)


def g11_weights(name):
    kw = G11_NETS[name]
    return generic_state_dict(1100 + sorted(G11_NETS).index(name), alpha_bias=-0.5, **kw)


def g11_dnerf_weights():
    kw = dict(G11_DNERF)
    ct = kw.pop("input_ch_time")
    sd = generic_state_dict(1150, alpha_bias=-0.5, prefix="_occ.", **kw)
    rng = np.random.default_rng(1151)
    W, D, skips, cin = kw["W"], kw["D"], kw["skips"], kw["input_ch"]
    ins = [cin + ct] + [W + cin if i in skips else W for i in range(D - 1)]
    for i, k in enumerate(ins):
        sd[f"_time.{i}.weight"] = (rng.standard_normal((W, k)) * np.sqrt(2.0 / k)).astype(np.float32)
        sd[f"_time.{i}.bias"] = (rng.standard_normal(W) * 0.05).astype(np.float32)
    sd["_time_out.weight"] = (rng.standard_normal((3, W)) * np.sqrt(2.0 / W) * 0.05).astype(np.float32)
    sd["_time_out.bias"] = (rng.standard_normal(3) * 0.01).astype(np.float32)
    return sd


def g11_inputs(n=300):
    rng = np.random.default_rng(111)
    pts = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    dirs = rng.standard_normal((n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    return dict(pts=pts, dirs=dirs.astype(np.float32), rays=g7_inputs(n=64, seed=111))


# ------------------------------------------------------------------ G12 render_rays without view directions, non-degenerate weights
G12_NET = dict(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)


def g12_weights():
    """(coarse, fine) nets of the shape `--use_viewdirs` unset gives (nerf/run.py:226-231); opacity biases tuned with the CPU
    oracle so that acc spans (0.2, 1.0) / (0.1, 1.0) on the G12 rays."""
    return synth.noview_state_dict(20250321, alpha_bias=0.5), synth.noview_state_dict(20250322, alpha_bias=0.7)


def g12_inputs(n=256):
    return g7_inputs(n=n, seed=57)
