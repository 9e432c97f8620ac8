#!/usr/bin/env python3
"""Golden vectors for the eval dataset counterpart: builds a tiny synthetic DTU-like dataset on
disk (seeded), runs the REFERENCE's datasets/dataloader_eval.MVSDataset on it and stores its
outputs.  Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_dataset_golden.py"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(HERE))  # tests/
sys.path.insert(0, os.environ.get("MVS_REFERENCE", "/root/reference"))
from synthetic_dataset import write_synthetic_dataset  # noqa: E402

from datasets.dataloader_eval import MVSDataset  # noqa: E402  (reference)


def main():
    out = {}
    with tempfile.TemporaryDirectory() as d:
        listfile = write_synthetic_dataset(d)
        ds = MVSDataset(os.path.join(d, "data"), listfile, "test", 3, 16, 1.06, pairfile="pair.txt",
                        cam_subfolder="Cameras", img_subfolder="Rectified/{}/rect_{:0>3}_3_r5000.png",
                        img_res=(96, 128), dataset_name="dtu")
        out["len"] = np.int64(len(ds))
        for idx in (0, 3, 5):
            s = ds[idx]
            out[f"{idx}_imgs"] = s["imgs"].astype(np.float32)
            out[f"{idx}_proj"] = s["proj_matrices"].astype(np.float32)
            out[f"{idx}_dv"] = s["depth_values"]
            out[f"{idx}_intr"] = np.stack(s["intrinsics"])
            out[f"{idx}_extr"] = np.stack(s["extrinsics"])
            out[f"{idx}_filename"] = np.array(s["filename"])
    np.savez_compressed(os.path.join(HERE, "fx_dataset.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
