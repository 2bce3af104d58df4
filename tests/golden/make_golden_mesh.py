#!/usr/bin/env python3
"""Capture g10_mesh_query.npz from the UNMODIFIED reference: nerf/extract_mesh.py's generate_viewdirs
and sample_grid, driven by the reference model through the reference's own 2-D run_network
(nerf/load_model.py:56-74), exactly as extract_mesh.main() wires them (:151-175).  skimage / trimesh /
configargparse / imageio are absent offline and not on this path: empty stand-in modules."""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

for name in ["skimage", "skimage.measure", "trimesh", "configargparse", "imageio", "lpips", "skimage.metrics", "cv2"]:
    try:
        __import__(name)
    except Exception:
        sys.modules[name] = types.ModuleType(name)
sys.modules["skimage"].measure = sys.modules["skimage.measure"]
for attr in ("peak_signal_noise_ratio", "structural_similarity"):
    if not hasattr(sys.modules["skimage.metrics"], attr):
        setattr(sys.modules["skimage.metrics"], attr, None)
REF = "/root/reference"
import torch  # noqa: E402
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "nerf"))


def load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


EM = load("nerf/extract_mesh.py", "ref_extract_mesh")
LM = load("nerf/load_model.py", "ref_load_model")
import embedder as EMB  # noqa: E402
import model as MODEL   # noqa: E402


@torch.no_grad()
def main():
    _, sd_f = cases.weights_static()
    net = MODEL.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_f.items()})
    embed_fn, _ = EMB.get_embedder(10, 3, 0)
    embeddirs_fn, _ = EMB.get_embedder(4, 3, 0)
    q = lambda inputs, viewdirs, fn: LM.run_network(inputs, viewdirs, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=65536)

    def batch_query_fn(positions, viewdirs):          # extract_mesh.py:155-175, device = cpu
        positions = torch.tensor(positions, dtype=torch.float32)
        viewdirs = torch.tensor(viewdirs, dtype=torch.float32)
        outputs = q(positions, viewdirs, net)
        if len(outputs.shape) == 3:
            outputs = outputs.squeeze(1)
        rgb = outputs[..., :3].numpy()
        return rgb[..., 0], rgb[..., 1], rgb[..., 2], outputs[..., 3].numpy()

    dens, col, (X, Y, Z) = EM.sample_grid(cases.G10_BOUNDS, cases.G10_RES, batch_query_fn, num_views=cases.G10_VIEWS, batch_size=50)
    vd = EM.generate_viewdirs(100)
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1)[:40]
    one = q(torch.tensor(pts, dtype=torch.float32), torch.tensor(np.tile(vd[3][None], (40, 1)), dtype=torch.float32), net)
    np.savez_compressed(os.path.join(HERE, "g10_mesh_query.npz"), density=dens, color=col, viewdirs100=vd,
                        per_point_raw=one.squeeze(1).numpy(), X=X)
    print("g10_mesh_query.npz", os.path.getsize(os.path.join(HERE, "g10_mesh_query.npz")) // 1024, "KiB",
          "density range", dens.min(), dens.max())


if __name__ == "__main__":
    main()
