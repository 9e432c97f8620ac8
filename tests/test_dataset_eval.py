"""Eval dataset counterpart against vectors captured from the reference's
datasets/dataloader_eval.MVSDataset on the same seeded synthetic dataset
(tests/golden/gen_dataset_golden.py -> tests/golden/fx_dataset.npz)."""
import os

import numpy as np
import pytest

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


def test_image_cache_gives_identical_samples(tmp_path):
    """EvalDataset(cache_images=N): decoded views are reused between samples -- same arrays, same
    adjusted intrinsics as the uncached (reference-like) path."""
    from synthetic_dataset import write_synthetic_dataset
    from scene_3dreconstruction_mvsnet_amd.dataset_eval import EvalDataset
    listfile = write_synthetic_dataset(str(tmp_path))
    kw = dict(mode="test", nviews=3, ndepths=16, interval_scale=1.06, img_res=(96, 128), dataset_name="dtu")
    import os
    plain = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, **kw)
    cached = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, cache_images=3, **kw)
    for rounds in range(2):
        for i in range(len(plain)):
            a, b = plain[i], cached[i]
            assert a["filename"] == b["filename"]
            for k in ("imgs", "proj_matrices", "depth_values"):
                np.testing.assert_array_equal(a[k], b[k])
            for x, y in zip(a["intrinsics"] + a["extrinsics"], b["intrinsics"] + b["extrinsics"]):
                np.testing.assert_array_equal(x, y)
    assert 0 < len(cached._img_cache) <= 3


def test_uint8_items_are_the_reference_items_before_the_division(tmp_path):
    """EvalDataset(image_dtype="uint8"): `imgs` are the decoded 8-bit pixels; the reference's conversion
    (np.array(img, float32) / 255., datasets/data_io.py:143) applied to them afterwards gives the float item BIT FOR BIT
    -- which is what the drop-in MVSNet does on the device.  Everything else of the item is unchanged, and the
    shared-memory decoder ring ships the uint8 block as it is (a quarter of the bytes)."""
    import os
    from synthetic_dataset import write_synthetic_dataset
    from scene_3dreconstruction_mvsnet_amd.dataset_eval import EvalDataset
    from scene_3dreconstruction_mvsnet_amd.decoder_pool import DecoderPool
    listfile = write_synthetic_dataset(str(tmp_path))
    kw = dict(mode="test", nviews=3, ndepths=16, interval_scale=1.06, img_res=(96, 128), dataset_name="dtu")
    f32 = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, **kw)
    u8 = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, image_dtype="uint8", cache_images=2, **kw)
    for i in range(len(f32)):
        a, b = f32[i], u8[i]
        assert b["imgs"].dtype == np.uint8 and b["imgs"].shape == a["imgs"].shape
        np.testing.assert_array_equal(b["imgs"].astype(np.float32) / np.float32(255.0), a["imgs"])
        np.testing.assert_array_equal(a["proj_matrices"], b["proj_matrices"])
        np.testing.assert_array_equal(a["depth_values"], b["depth_values"])
        assert a["filename"] == b["filename"]
    with DecoderPool(u8, procs=2, chunk=2, slots=4) as pool:
        for i, s in zip(range(len(u8)), pool.imap(list(range(len(u8))))):
            assert np.array(s["imgs"]).dtype == np.uint8 and np.array(s["imgs"]).nbytes * 4 == f32[i]["imgs"].nbytes
            np.testing.assert_array_equal(np.array(s["imgs"]), u8[i]["imgs"])
            pool.release(s)
    with pytest.raises(ValueError):
        EvalDataset(os.path.join(str(tmp_path), "data"), listfile, image_dtype="float16", **kw)


def test_decoder_pool_delivers_the_dataset_in_order(tmp_path):
    """Worker processes + shared-memory ring: every sample equals dataset[i], in order, also when
    the consumer never releases explicitly and when there are more samples than slots."""
    import os
    from synthetic_dataset import write_synthetic_dataset
    from scene_3dreconstruction_mvsnet_amd.dataset_eval import EvalDataset
    from scene_3dreconstruction_mvsnet_amd.decoder_pool import DecoderPool
    listfile = write_synthetic_dataset(str(tmp_path))
    ds = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, "test", 3, 16, 1.06, img_res=(96, 128),
                     dataset_name="dtu", cache_images=4)
    order = list(range(len(ds))) * 3          # 24 samples through 4 slots
    with DecoderPool(ds, procs=2, chunk=2, slots=4) as pool:
        seg = pool  # the segment name is only known after the first item
        for explicit in (True, False):
            n = 0
            for i, s in zip(order, pool.imap(order)):
                want = ds[i]
                assert s["filename"] == want["filename"]
                np.testing.assert_array_equal(np.array(s["imgs"]), want["imgs"])
                np.testing.assert_array_equal(s["proj_matrices"], want["proj_matrices"])
                np.testing.assert_array_equal(s["depth_values"], want["depth_values"])
                if explicit:
                    pool.release(s)
                n += 1
            assert n == len(order)
        name = seg._shm.name
    assert not os.path.exists(os.path.join("/dev/shm", name.lstrip("/")))   # unlinked on close


def test_decoder_pool_forwards_worker_errors(tmp_path):
    from scene_3dreconstruction_mvsnet_amd.decoder_pool import DecoderPool

    with DecoderPool(_FailingDataset(), procs=1, chunk=1, slots=2, slot_bytes=4096) as pool:
        it = pool.imap([0, 1, 2])
        next(it)
        with pytest.raises(RuntimeError, match="item 1 failed"):
            next(it)


class _FailingDataset:
    def __getitem__(self, i):
        if i == 1:
            raise KeyError("broken sample")
        return {"imgs": np.zeros((1, 3, 4, 4), np.float32), "filename": str(i)}


def test_view_decoder_pool_decodes_every_view_once_and_matches_the_dataset(tmp_path):
    """View-level pool: samples assembled from a shared-memory cache of decoded views equal
    dataset[i] bit for bit, also with a cache smaller than the lookahead window wants."""
    import os
    from synthetic_dataset import write_synthetic_dataset
    from scene_3dreconstruction_mvsnet_amd.dataset_eval import EvalDataset
    from scene_3dreconstruction_mvsnet_amd.decoder_pool import ViewDecoderPool
    listfile = write_synthetic_dataset(str(tmp_path))
    ds = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, "test", 3, 16, 1.06, img_res=(96, 128),
                     dataset_name="dtu")
    # the three-step form of a sample is the sample
    for i in range(len(ds)):
        want = ds[i]
        filename, views = ds.view_plan(i)
        dec = [ds.decode_view(p) for p, _ in views]
        got = ds.assemble(i, [a for _, a in dec])
        assert filename == want["filename"]
        np.testing.assert_array_equal(np.stack([x for x, _ in dec]), want["imgs"])
        np.testing.assert_array_equal(got["proj_matrices"], want["proj_matrices"])
    order = list(range(len(ds))) * 3
    with ViewDecoderPool(ds, procs=2, slots=5, lookahead=3) as pool:
        for explicit in (True, False):
            n = 0
            for i, s in zip(order, pool.imap(order)):
                want = ds[i]
                assert s["filename"] == want["filename"] and isinstance(s["imgs"], list)
                np.testing.assert_array_equal(np.stack(s["imgs"]), want["imgs"])
                np.testing.assert_array_equal(s["proj_matrices"], want["proj_matrices"])
                np.testing.assert_array_equal(s["depth_values"], want["depth_values"])
                if explicit:
                    pool.release(s)
                n += 1
            assert n == len(order)
        name = pool._shm.name
    assert not os.path.exists(os.path.join("/dev/shm", name.lstrip("/")))
    with ViewDecoderPool(ds, procs=1, slots=2, lookahead=2) as small:       # fewer slots than one sample's views
        with pytest.raises(RuntimeError, match="slots"):
            next(small.imap([0]))
