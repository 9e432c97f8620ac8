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


def save_pfm(filename: str, image: np.ndarray, scale: float = 1) -> None:
    """Same bytes as reference datasets/data_io.py:46-73 (float32 only, like the reference)."""
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    image = np.flipud(image)
    if len(image.shape) == 3 and image.shape[2] == 3:
        color = True
    elif len(image.shape) == 2 or (len(image.shape) == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    endian = image.dtype.byteorder
    if endian == "<" or (endian == "=" and sys.byteorder == "little"):
        scale = -scale
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write("{} {}\n".format(image.shape[1], image.shape[0]).encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        image.tofile(f)


def read_pfm(filename: str):
    """Inverse of save_pfm (reference datasets/data_io.py:8-43): returns (data, scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header == "PF":
            color = True
        elif header == "Pf":
            color = False
        else:
            raise Exception("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = map(int, m.groups())
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        scale = abs(scale)
        data = np.fromfile(f, endian + "f")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape)), scale


def depth_map_paths(outdir: str, filename_template: str):
    """`filename = '{scan}/{{}}/{view:08d}{{}}'` (datasets/dataloader_eval.py:176) ->
    (depth .pfm, confidence .pfm) paths as eval.py:377-392 builds them."""
    import os
    return (os.path.join(outdir, filename_template.format("depth_est", ".pfm")),
            os.path.join(outdir, filename_template.format("confidence", ".pfm")))
