"""Minimal PNG writer (zlib + struct) for the output side of render_path: the reference calls
`imageio.imwrite(filename, to8b(rgb))` (nerf/run.py:210-213, d_nerf/run_dnerf.py:222-230); imageio is not a
dependency of this package.  8-bit grey, grey+alpha, RGB or RGBA, no interlacing, filter type 0."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def write_png(filename, img):
    """img: uint8 array [H,W], [H,W,1|2|3|4]."""
    a = np.asarray(img)
    if a.dtype != np.uint8:
        raise TypeError(f"write_png expects uint8 (use to8b), got {a.dtype}")
    if a.ndim == 2:
        a = a[..., None]
    if a.ndim != 3 or a.shape[2] not in (1, 2, 3, 4) or a.shape[0] == 0 or a.shape[1] == 0:
        raise ValueError(f"write_png: unsupported image shape {a.shape}")
    h, w, c = a.shape
    color_type = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), np.ascontiguousarray(a).reshape(h, w * c)], axis=1).tobytes()
    png = (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
           + _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b""))
    with open(filename, "wb") as f:
        f.write(png)
