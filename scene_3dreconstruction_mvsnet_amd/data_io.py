"""PFM codec with the byte format of the reference's datasets/data_io.py (eval.py output contract).

eval.py:387,392 writes `outputs["depth"][b]` and `outputs["photometric_confidence"][b]` with
`save_pfm` (datasets/data_io.py:46-73): ASCII header `Pf\\n{w} {h}\\n-1.000000\\n` (negative scale
= little endian) followed by h*w little-endian float32, rows bottom-to-top.  `read_pfm`
(datasets/data_io.py:8-43) is the inverse.  Provided so that a sharded eval driver can write the
reference's file tree without importing the reference ("next" row f1).
"""
from __future__ import annotations

import re
import sys

import numpy as np


_PFM_MAGIC = {1: b"Pf", 3: b"PF"}   # channels -> magic (grey / colour)


def _pfm_channels(shape) -> int:
    if len(shape) == 2 or (len(shape) == 3 and shape[2] == 1):
        return 1
    if len(shape) == 3 and shape[2] == 3:
        return 3
    raise ValueError("PFM holds H x W, H x W x 1 or H x W x 3 arrays, got shape %r" % (tuple(shape),))


def save_pfm(filename: str, image: np.ndarray, scale: float = 1) -> None:
    """Write `image` as PFM: magic line, "W H" line, scale line whose SIGN carries the byte order
    (negative = little endian), then the float32 rows bottom-to-top.  Byte-for-byte what the
    reference's writer produces (datasets/data_io.py:46-73); float32 input only, like the reference."""
    if image.dtype != np.float32:
        raise TypeError("save_pfm takes float32 arrays (got %s)" % image.dtype)
    channels = _pfm_channels(image.shape)
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    signed_scale = -float(scale) if little else float(scale)
    height, width = image.shape[0], image.shape[1]
    header = b"%s\n%d %d\n%s\n" % (_PFM_MAGIC[channels], width, height, ("%f" % signed_scale).encode("ascii"))
    with open(filename, "wb") as f:
        f.write(header)
        f.write(np.ascontiguousarray(image[::-1]).tobytes())   # last row first


def read_pfm(filename: str):
    """-> (array [H,W] or [H,W,3] float32 in top-to-bottom row order, |scale|); inverse of save_pfm
    (the reference's reader: datasets/data_io.py:8-43)."""
    with open(filename, "rb") as f:
        magic = f.readline().strip()
        channels = {v: k for k, v in _PFM_MAGIC.items()}.get(magic)
        if channels is None:
            raise ValueError("%s: not a PFM file (magic %r)" % (filename, magic))
        dims = re.fullmatch(rb"(\d+)\s(\d+)\s", f.readline())
        if dims is None:
            raise ValueError("%s: malformed PFM size line" % filename)
        width, height = int(dims.group(1)), int(dims.group(2))
        scale = float(f.readline().strip())
        data = np.fromfile(f, dtype=("<f4" if scale < 0 else ">f4"))
    rows = data.reshape((height, width, 3) if channels == 3 else (height, width))
    return rows[::-1], abs(scale)


def depth_map_paths(outdir: str, filename_template: str):
    """`filename = '{scan}/{{}}/{view:08d}{{}}'` (datasets/dataloader_eval.py:176) ->
    (depth .pfm, confidence .pfm) paths as eval.py:377-392 builds them."""
    import os
    return (os.path.join(outdir, filename_template.format("depth_est", ".pfm")),
            os.path.join(outdir, filename_template.format("confidence", ".pfm")))
