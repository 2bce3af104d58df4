"""Camera / pose front-end of the render path: the step
immediately before `render()`.  Pure host numpy, restated from the reference's data loaders
WITHOUT the image I/O: what they compute from `transforms_*.json` / `poses_bounds.npy` to
produce (poses, render_poses, hwf, near/far).

  blender : dataloader/load_blender.py:11-35 (pose_spherical), :130-146 (focal, render path, half_res)
  llff    : dataloader/load_llff.py:61-66,102-104 (poses_bounds layout), :120-179 (normalize, viewmatrix,
            poses_avg, render_path_spiral, recenter_poses), :244-317 (load_llff_data pipeline)
Pinned by tests/golden/g9_cameras.npz (captured by running the reference loaders on a synthetic
on-disk dataset, tests/golden/make_golden_cameras.py)."""
import json
import os

import numpy as np

from .synth import pose_spherical  # noqa: F401  (load_blender.py:30-35)


def blender_render_poses(n=360, phi=-30.0, radius=4.0):
    """The 360-view orbit of load_blender.py:136 as float32 [n,4,4] (n+1 linspace points, last dropped)."""
    return np.stack([pose_spherical(a, phi, radius) for a in np.linspace(-180, 180, n + 1)[:-1]], 0)


def blender_hwf(H, W, camera_angle_x, half_res=False):
    """load_blender.py:132-141: focal from the horizontal FOV; half_res halves H, W, focal."""
    focal = .5 * W / np.tan(.5 * float(camera_angle_x))
    if half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
    return [H, W, focal]


def blender_meta(basedir, testskip=1):
    """Poses + split indices + camera_angle_x from transforms_{train,val,test}.json (load_blender.py:82-127),
    without reading the PNGs."""
    poses, counts, angle = [], [0], None
    for s in ("train", "val", "test"):
        with open(os.path.join(basedir, f"transforms_{s}.json")) as fp:
            meta = json.load(fp)
        skip = 1 if (s == "train" or testskip == 0) else testskip
        p = [np.array(f["transform_matrix"]) for f in meta["frames"][::skip]]
        poses.append(np.array(p).astype(np.float32))
        counts.append(counts[-1] + len(p))
        angle = float(meta["camera_angle_x"])
    return np.concatenate(poses, 0), [np.arange(counts[i], counts[i + 1]) for i in range(3)], angle


def intrinsics(H, W, focal):
    """nerf/run.py:518-523."""
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])


# ---------------------------------------------------------------------------------- LLFF
def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """load_llff.py:123-129: columns [right, up', forward, position]."""
    fwd = normalize(z)
    right = normalize(np.cross(up, fwd))
    up2 = normalize(np.cross(fwd, right))
    return np.stack([right, up2, fwd, pos], 1)


def poses_avg(poses):
    """load_llff.py:135-145 -> [3,5] (average camera, hwf column carried along)."""
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    fwd = normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(fwd, up, center), hwf], 1)


def recenter_poses(poses):
    """load_llff.py:162-175: express every pose in the frame of the average camera."""
    out = poses + 0
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    c2w = np.concatenate([poses_avg(poses)[:3, :4], bottom], -2)
    full = np.concatenate([poses[:, :3, :4], np.tile(bottom[None], [poses.shape[0], 1, 1])], -2)
    full = np.linalg.inv(c2w) @ full
    out[:, :3, :4] = full[:, :3, :4]
    return out


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """load_llff.py:149-158."""
    out = []
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    for theta in np.linspace(0., 2. * np.pi * rots, N + 1)[:-1]:
        c = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        z = normalize(c - np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.])))
        out.append(np.concatenate([viewmatrix(z, up, c), hwf], 1))
    return out


def llff_from_poses_bounds(poses_arr, image_hw, factor=8, recenter=True, bd_factor=.75, path_zflat=False):
    """Everything load_llff_data (load_llff.py:244-317) derives from `poses_bounds.npy` ([N,17]) and the
    size of the (already down-scaled) images: returns (poses [N,3,5] f32, bds [N,2] f32,
    render_poses [120,3,5] f32, i_test).  `spherify` is not restated (its body is missing in the reference)."""
    poses = poses_arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])
    bds = poses_arr[:, -2:].transpose([1, 0])
    poses = poses.copy()
    poses[:2, 4, :] = np.array(image_hw).reshape([2, 1])                # load_llff.py:102-103
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)
    c2w = poses_avg(poses)
    up = normalize(poses[:, :3, 1].sum(0))
    close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
    dt = .75
    focal = 1. / (((1. - dt) / close_depth + dt / inf_depth))
    zdelta = close_depth * .2
    rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
    c2w_path, n_views, n_rots = c2w, 120, 2
    if path_zflat:
        zloc = -close_depth * .1
        c2w_path[:3, 3] = c2w_path[:3, 3] + zloc * c2w_path[:3, 2]
        rads[2] = 0.
        n_rots, n_views = 1, 60
    render_poses = np.array(render_path_spiral(c2w_path, up, rads, focal, zdelta, zrate=.5, rots=n_rots, N=n_views)).astype(np.float32)
    c2w = poses_avg(poses)
    i_test = int(np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1)))
    return poses.astype(np.float32), bds, render_poses, i_test


def llff_near_far(bds, no_ndc=False):
    """nerf/run.py:451-458."""
    if no_ndc:
        return float(np.ndarray.min(bds) * .9), float(np.ndarray.max(bds) * 1.)
    return 0., 1.
