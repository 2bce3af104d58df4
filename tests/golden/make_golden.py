#!/usr/bin/env python3
"""Capture golden vectors G1-G8 from the UNMODIFIED reference (build container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; ~1 min on 8 cores)

The reference's math modules (ray.py, embedder.py, model.py) import as they are; the two
runner modules import I/O-only packages that are absent offline (imageio, lpips,
skimage, cv2, configargparse, tensorboard) - these are replaced by EMPTY module
objects in sys.modules; nothing on the hot path touches them.
Only OUTPUTS of the reference (plus a checksum of the seeded inputs of cases.py) are
written, as small float32 .npz files; no reference source text is stored.
"""
import os
import sys
import types
import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

REF = "/root/reference"
import importlib
for name in ["imageio", "lpips", "skimage", "skimage.metrics", "cv2", "configargparse",
             "torch.utils.tensorboard"]:
    try:
        importlib.import_module(name)
    except Exception:
        sys.modules[name] = types.ModuleType(name)
for attr in ("peak_signal_noise_ratio", "structural_similarity"):
    if not hasattr(sys.modules["skimage.metrics"], attr):
        setattr(sys.modules["skimage.metrics"], attr, None)
if not hasattr(sys.modules["torch.utils.tensorboard"], "SummaryWriter"):
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object

import torch  # noqa: E402
sys.path.insert(0, REF)
import ray as RAY          # noqa: E402
import embedder as EMB     # noqa: E402
import model as MODEL      # noqa: E402
import importlib.util      # noqa: E402


def _load(path, name):
    cwd = os.getcwd()
    os.chdir(os.path.dirname(path))
    sys.path.insert(0, os.path.dirname(path))
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        os.chdir(cwd)
        sys.path.pop(0)
    return mod


RUN = _load(os.path.join(REF, "nerf", "run.py"), "ref_nerf_run")
DRUN = _load(os.path.join(REF, "d_nerf", "run_dnerf.py"), "ref_dnerf_run")
torch.set_default_dtype(torch.float32)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
N = lambda t: t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def load_sd(module, sd_np):
    module.load_state_dict({k: T(v) for k, v in sd_np.items()}, strict=True)
    return module.eval()


@torch.no_grad()
def main():
    # ---- G1
    g = cases.g1_inputs()
    e10, d10 = EMB.get_embedder(10, 3, 0)
    e4, d4 = EMB.get_embedder(4, 3, 0)
    et, dt = EMB.get_embedder(10, 1, 0)
    assert (d10, d4, dt) == (63, 27, 21)
    save("g1_embed", pts=N(e10(T(g["pts"]))), dirs=N(e4(T(g["dirs"]))), t=N(et(T(g["t"]))),
         crc=cases.checksum(g["pts"], g["dirs"], g["t"]))

    # ---- G2
    g = cases.g2_inputs()
    o_k, d_k = RAY.get_rays(400, 400, g["K400"], T(g["c2w400"]))
    o_f, d_f = RAY.get_rays(400, 400, g["focal400"], T(g["c2w400"]))
    o_s, d_s = RAY.get_rays(32, 48, g["K_small"], T(g["c2w_small"]))
    on, dn = RAY.get_rays_np(400, 400, g["K400"], g["c2w400"])
    Kf = g["Kf"]
    o_l, d_l = RAY.get_rays(378, 504, Kf, T(g["c2wf"]))
    o_ndc, d_ndc = RAY.ndc_rays(378, 504, Kf[0][0], 1., o_l, d_l)
    step = 37   # subsample the big grids; the small grid is stored whole
    flat = lambda t: N(t).reshape(-1, 3)
    save("g2_rays", d_k=flat(d_k)[::step], o_k=flat(o_k)[::step], d_f=flat(d_f)[::step],
         d_s=flat(d_s), o_s=flat(o_s), d_np=on.reshape(-1, 3)[::step] * 0 + dn.reshape(-1, 3)[::step],
         np_dtype=np.array([str(dn.dtype)]), d_l=flat(d_l)[::step],
         o_ndc=flat(o_ndc)[::step], d_ndc=flat(d_ndc)[::step], step=np.array([step]),
         crc=cases.checksum(g["K400"], g["c2w400"], g["K_small"], g["c2w_small"], g["Kf"], g["c2wf"]))

    # ---- G3 (coarse z / pts exactly as nerf/run.py:355-385 does them, via render_rays with a probe net)
    g = cases.g3_inputs()
    probe = {}

    def probe_query(pts, viewdirs, fn):
        probe["pts"] = pts.clone()
        return torch.zeros(list(pts.shape[:-1]) + [4])
    out = {}
    for lindisp in (False, True):
        for perturb in (0., 1.):
            rb = torch.cat([T(g["rays_o"]), T(g["rays_d"]), T(g["near"]), T(g["far"]),
                            T(g["rays_d"]) / torch.norm(T(g["rays_d"]), dim=-1, keepdim=True)], -1)
            RUN.render_rays(rb, None, probe_query, 64, lindisp=lindisp, perturb=perturb, pytest=True)
            out[f"pts_l{int(lindisp)}_p{int(perturb)}"] = N(probe["pts"])[:32]
    save("g3_coarse", crc=cases.checksum(g["rays_o"], g["rays_d"], g["near"], g["far"], g["t_rand"]), **out)

    # ---- G4
    g = cases.g4_inputs()
    sd_c, sd_f = cases.weights_static()
    sd_d = cases.weights_dnerf()
    van = load_sd(MODEL.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5,
                                     skips=[4], use_viewdirs=True), sd_c)
    x = T(g["x"])
    y_van = van(x)
    orig = load_sd(MODEL.NeRFOriginal(D=8, W=256, input_ch=63, input_ch_views=27, input_ch_time=21,
                                      output_ch=5, skips=[4], use_viewdirs=True, embed_fn=e10), sd_f)
    y_orig, z_orig = orig(x, None)
    assert float(z_orig.abs().max()) == 0.0 and tuple(z_orig.shape) == (4096, 3)
    dn = load_sd(MODEL.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=63, output_ch=5,
                                        skips=[4], input_ch_views=27, input_ch_time=21,
                                        use_viewdirs=True, embed_fn=e10, zero_canonical=True), sd_d)
    res = {}
    for tv in (0.0, 0.5):
        te = et(torch.full((4096, 1), tv))
        y, dx = dn(x, [te, te])
        res[f"dn_out_t{int(tv*10)}"] = N(y)
        res[f"dn_dx_t{int(tv*10)}"] = N(dx)
    save("g4_mlp", vanilla=N(y_van), original=N(y_orig), crc=cases.checksum(g["x"]), **res)

    # ---- G5
    for S in (64, 192):
        g = cases.g5_inputs(S)
        res = {}
        for wb in (False, True):
            r = RAY.raw2outputs(T(g["raw"]), T(g["z"]), T(g["rays_d"]), 0, wb)
            for k, v in zip(["rgb", "disp", "acc", "weights", "depth"], r):
                res[f"{k}_w{int(wb)}"] = N(v)
        r = RAY.raw2outputs(T(g["raw"]), T(g["z"]), T(g["rays_d"]), 1.0, True, pytest=True)
        for k, v in zip(["rgb", "disp", "acc", "weights", "depth"], r):
            res[f"{k}_noise"] = N(v)
        save(f"g5_raw2outputs_S{S}", crc=cases.checksum(g["raw"], g["z"], g["rays_d"], g["noise"]), **res)

    # ---- G6
    g = cases.g6_inputs()
    s_det = RAY.sample_pdf(T(g["bins"]), T(g["weights"]), 128, det=True)
    s_rnd = RAY.sample_pdf(T(g["bins"]), T(g["weights"]), 128, det=False, pytest=True)
    zs_det, _ = torch.sort(torch.cat([T(g["z"]), s_det], -1), -1)
    zs_rnd, _ = torch.sort(torch.cat([T(g["z"]), s_rnd], -1), -1)
    save("g6_sample_pdf", det=N(s_det), rnd=N(s_rnd), z_det=N(zs_det), z_rnd=N(zs_rnd),
         std_det=N(torch.std(s_det, dim=-1, unbiased=False)), std_rnd=N(torch.std(s_rnd, dim=-1, unbiased=False)),
         crc=cases.checksum(g["z"], g["bins"], g["weights"], g["u"]))

    # ---- G7 static render_rays / render
    fine = load_sd(MODEL.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5,
                                      skips=[4], use_viewdirs=True), sd_f)
    q = lambda inputs, viewdirs, network_fn: RUN.run_network(
        inputs, viewdirs, network_fn, embed_fn=e10, embeddirs_fn=e4, netchunk=1024 * 64)

    def keep(ret, nraw=32):
        o = {}
        for k, v in ret.items():
            v = N(v)
            o[k] = v[:nraw] if k in ("raw", "position_delta", "position_delta_0") else v
        return o

    def ray_batch(g, extra=None):
        o, d = T(g["rays_o"]), T(g["rays_d"])
        cols = [o, d, g["near"] * torch.ones_like(d[:, :1]), g["far"] * torch.ones_like(d[:, :1])]
        if extra is not None:
            cols.append(extra * torch.ones_like(d[:, :1]))
        cols.append(d / torch.norm(d, dim=-1, keepdim=True))
        return torch.cat(cols, -1)

    g = cases.g7_inputs()
    rb = ray_batch(g)
    r = RUN.render_rays(rb, van, q, 64, retraw=True, N_importance=0, white_bkgd=True)
    acc = N(r["acc_map"])
    print("C1 acc: mean %.3f min %.3f max %.3f" % (acc.mean(), acc.min(), acc.max()))
    save("g7_c1", crc=cases.checksum(g["rays_o"], g["rays_d"]), **keep(r))
    r = RUN.render_rays(rb, van, q, 64, retraw=True, N_importance=128, network_fine=fine, white_bkgd=True)
    acc = N(r["acc_map"])
    print("C2 acc: mean %.3f min %.3f max %.3f" % (acc.mean(), acc.min(), acc.max()))
    save("g7_c2", crc=cases.checksum(g["rays_o"], g["rays_d"]), **keep(r))
    # the same through render(): rays given, use_viewdirs, no ndc (nerf/run.py:105-169)
    K, _ = cases.synth.lego_camera(400, 400)
    rr = RUN.render(400, 400, K, chunk=1024 * 32, rays=(T(g["rays_o"]), T(g["rays_d"])), ndc=False,
                    near=2., far=6., use_viewdirs=True, network_fn=van, network_query_fn=q, N_samples=64,
                    N_importance=128, network_fine=fine, white_bkgd=True, perturb=0., raw_noise_std=0.)
    assert np.array_equal(N(rr[0]), N(r["rgb_map"]))
    # lindisp + no white bkgd + single network for the fine pass
    gs = cases.g7_inputs(n=256, seed=11)
    r = RUN.render_rays(ray_batch(gs), van, q, 64, retraw=False, N_importance=128, network_fine=None,
                        white_bkgd=False, lindisp=True)
    save("g7_lindisp", crc=cases.checksum(gs["rays_o"], gs["rays_d"]), **keep(r))
    # perturb=1 through the reference's own pytest hook
    r = RUN.render_rays(ray_batch(gs), van, q, 64, retraw=False, N_importance=128, network_fine=fine,
                        white_bkgd=True, perturb=1., pytest=True)
    save("g7_perturb", crc=cases.checksum(gs["rays_o"], gs["rays_d"]), **keep(r))
    # NDC (fern-like) through render()
    gn = cases.g7_ndc_inputs()
    Kf, _ = cases.synth.fern_camera()
    rr = RUN.render(378, 504, Kf, chunk=1024 * 32, rays=(T(gn["rays_o"]), T(gn["rays_d"])), ndc=True,
                    near=0., far=1., use_viewdirs=True, network_fn=van, network_query_fn=q, N_samples=64,
                    N_importance=128, network_fine=fine, white_bkgd=False, perturb=0., raw_noise_std=0.)
    save("g7_ndc", rgb_map=N(rr[0]), disp_map=N(rr[1]), acc_map=N(rr[2]),
         crc=cases.checksum(gn["rays_o"], gn["rays_d"]), **{k: N(v) for k, v in rr[3].items()})

    # ---- G8 D-NeRF render_rays
    qd = lambda inputs, viewdirs, ts, network_fn: DRUN.run_network(
        inputs, viewdirs, ts, network_fn, embed_fn=e10, embeddirs_fn=e4, embedtime_fn=et,
        netchunk=1024 * 64, embd_time_discr=True)
    g = cases.g8_inputs()
    for tv in (0.0, 0.5):
        rb = ray_batch(g, extra=tv)
        r = DRUN.render_rays(rb, dn, qd, 64, retraw=True, N_importance=128, white_bkgd=True)
        acc = N(r["acc_map"])
        dxm = np.abs(N(r["position_delta"])).max()
        print("C5 t=%.1f acc: mean %.3f min %.3f max %.3f  |dx|max %.4f" % (tv, acc.mean(), acc.min(), acc.max(), dxm))
        save(f"g8_dnerf_t{int(tv*10)}", crc=cases.checksum(g["rays_o"], g["rays_d"]), **keep(r))
    r = DRUN.render_rays(ray_batch(g, extra=0.25)[:128], dn, qd, 64, retraw=False, N_importance=0, white_bkgd=True)
    save("g8_dnerf_coarse_only", crc=cases.checksum(g["rays_o"], g["rays_d"]), **keep(r))


if __name__ == "__main__":
    main()
