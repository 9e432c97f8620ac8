"""Eval dataset counterpart against vectors captured from the reference's
datasets/dataloader_eval.MVSDataset on the same seeded synthetic dataset
(tests/golden/gen_dataset_golden.py -> tests/golden/fx_dataset.npz)."""
import os

import numpy as np

from conftest import GOLDEN
from scene_3dreconstruction_mvsnet_amd.dataset_eval import EvalDataset, parse_pair_file
from synthetic_dataset import write_synthetic_dataset


def test_eval_dataset_matches_reference(tmp_path):
    listfile = write_synthetic_dataset(str(tmp_path))
    ds = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, "test", 3, 16, 1.06,
                     pairfile="pair.txt", cam_subfolder="Cameras",
                     img_subfolder="Rectified/{}/rect_{:0>3}_3_r5000.png", img_res=(96, 128),
                     dataset_name="dtu")
    with np.load(os.path.join(GOLDEN, "fx_dataset.npz")) as fx:
        assert len(ds) == int(fx["len"]) == 8
        for idx in (0, 3, 5):
            s = ds[idx]
            assert s["imgs"].shape == fx[f"{idx}_imgs"].shape == (3, 3, 96, 128)
            np.testing.assert_array_equal(s["imgs"].astype(np.float32), fx[f"{idx}_imgs"])
            np.testing.assert_array_equal(s["proj_matrices"].astype(np.float32), fx[f"{idx}_proj"])
            np.testing.assert_array_equal(s["depth_values"], fx[f"{idx}_dv"])
            np.testing.assert_array_equal(np.stack(s["intrinsics"]), fx[f"{idx}_intr"])
            np.testing.assert_array_equal(np.stack(s["extrinsics"]), fx[f"{idx}_extr"])
            assert s["filename"] == str(fx[f"{idx}_filename"])
            assert s["depth_values"].dtype == np.float32 and len(s["depth_values"]) == 16


def test_parse_pair_file(tmp_path):
    listfile = write_synthetic_dataset(str(tmp_path))
    pairs = parse_pair_file(os.path.join(os.path.dirname(listfile), "data", "pair.txt"))
    assert pairs[0] == (0, [1, 2, 3]) and pairs[3] == (3, [0, 1, 2])
