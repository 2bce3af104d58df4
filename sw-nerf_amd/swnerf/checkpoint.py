"""Checkpoint compatibility: the reference's `*.tar` files are plain
torch.save dicts (nerf/run.py:716-724, d_nerf/run_dnerf.py:757-770):
    global_step, network_fn_state_dict, [network_fine_state_dict], optimizer_state_dict, [amp]
Parameter names/shapes of the swnerf modules equal the reference's, so these load unchanged; this
module restates the discovery / load / save logic of create_nerf (nerf/run.py:261-280,
run_dnerf.py:304-324) and the output side of render_path (`to8b`, nerf/run.py:210)."""
import os

import numpy as np
import torch

to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


def find_checkpoints(basedir, expname, ft_path=None):
    """Sorted candidate files: an explicit `ft_path`, else every file containing 'tar' in basedir/expname."""
    if ft_path is not None and ft_path != 'None':
        return [ft_path]
    d = os.path.join(basedir, expname)
    if not os.path.isdir(d):
        return []
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f]


def load_checkpoint(path, network_fn, network_fine=None, optimizer=None, map_location=None):
    """Returns global_step.  Missing 'network_fine_state_dict' is an error only if a fine net is passed."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    network_fn.load_state_dict(ckpt['network_fn_state_dict'])
    if network_fine is not None:
        network_fine.load_state_dict(ckpt['network_fine_state_dict'])
    if optimizer is not None and ckpt.get('optimizer_state_dict'):       # absent or {} (saved without an optimizer): keep the fresh state
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return int(ckpt['global_step'])


def save_checkpoint(basedir, expname, step, global_step, network_fn, network_fine=None, optimizer=None):
    """Writes basedir/expname/{step:06d}.tar with the reference's keys; returns the path."""
    os.makedirs(os.path.join(basedir, expname), exist_ok=True)
    path = os.path.join(basedir, expname, '{:06d}.tar'.format(step))
    d = {'global_step': global_step, 'network_fn_state_dict': network_fn.state_dict()}
    if network_fine is not None:
        d['network_fine_state_dict'] = network_fine.state_dict()
    if optimizer is not None:
        d['optimizer_state_dict'] = optimizer.state_dict()
    torch.save(d, path)
    return path


def reload_latest(basedir, expname, network_fn, network_fine=None, optimizer=None, ft_path=None, no_reload=False,
                  map_location=None):
    """The reload block of create_nerf: newest checkpoint, or start = 0 when there is none / no_reload."""
    ckpts = find_checkpoints(basedir, expname, ft_path)
    if len(ckpts) > 0 and not no_reload:
        return load_checkpoint(ckpts[-1], network_fn, network_fine, optimizer, map_location), ckpts[-1]
    return 0, None
