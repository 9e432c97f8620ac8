"""Eval-time dataset with the semantics of the reference's datasets/dataloader_eval.py
(SURVEY.md §8 f2): it defines what `MVSNet.forward` is fed during `save_depth` (eval.py:292-305).

Written from the reference's behaviour, not its code:
  * pair file (dataloader_eval.py:41-49): first line = number of viewpoints; then per viewpoint a
    line with the reference view id and a line `n id0 score0 id1 score1 ...` (source ids are the
    odd-position tokens); the same pair file is applied to every scan of the list file
  * cam file (dataloader_eval.py:54-71): line 0 `extrinsic`, lines 1-4 a 4x4 matrix, line 6
    `intrinsic`, lines 7-9 a 3x3 matrix, line 11 `depth_min depth_interval ...`; the interval is
    multiplied by `interval_scale`
  * image (datasets/data_io.py:76-154): shrink with PIL bilinear by the larger of the two target
    ratios (never enlarge), centre-crop to the target (or to a multiple of 32 below it), intrinsics
    rows 0-1 scaled and principal point shifted accordingly, pixels / 255
  * sample (dataloader_eval.py:101-176): reference view + first nviews-1 sources; intrinsics rows
    0-1 divided by 4 (feature scale); proj = [[K @ E[:3,:4]], [E[3]]]; depth_values =
    arange(depth_min, interval*(D-0.5)+depth_min, interval) as float32 from the REFERENCE view's
    cam file; DTU image files are numbered from 1 (`vid + 1`)
"""
from __future__ import annotations

import math
import os

import numpy as np
from PIL import Image


def parse_pair_file(path: str):
    """-> [(ref_view, [src_views...]), ...]"""
    with open(path) as f:
        tokens = f.read().split("\n")
    n = int(tokens[0])
    out = []
    for i in range(n):
        ref = int(tokens[1 + 2 * i].strip())
        src = [int(t) for t in tokens[2 + 2 * i].split()[1::2]]
        out.append((ref, src))
    return out


def parse_cam_file(path: str, interval_scale: float = 1.0):
    """-> (intrinsics 3x3 f32, extrinsics 4x4 f32, depth_min, depth_interval * interval_scale)"""
    with open(path) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    extr = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intr = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    fields = lines[11].split()
    return intr, extr, float(fields[0]), float(fields[1]) * interval_scale


def load_image_rescaled_cropped(path: str, intrinsics: np.ndarray, img_res=(512, 640), base: int = 32,
                                cache: "dict | None" = None, cache_size: int = 0, image_dtype: str = "float32"):
    """-> (float32 HxWx3 in [0,1], adjusted intrinsics).  `intrinsics` is modified in place too,
    as the reference does.  image_dtype="uint8": the decoded pixels BEFORE the reference's
    `np.array(img, dtype=np.float32) / 255.` (datasets/data_io.py:143) -- the drop-in MVSNet divides on the device,
    bit-identically, and the host-to-device copy is a quarter of the bytes.

    `cache` (an insertion-ordered dict used as an LRU of at most `cache_size` entries) keeps the decoded,
    rescaled and cropped pixels per (path, img_res) together with the scale and crop offsets the
    intrinsics are adjusted by: in an eval run every view is decoded as the reference view of one
    sample and again as a source view of several neighbours.  The returned pixel array is then
    shared between samples (read-only use: np.stack copies it into the sample)."""
    if image_dtype not in ("float32", "uint8"):
        raise ValueError(f"image_dtype must be 'float32' or 'uint8', got {image_dtype!r}")
    key = (path, tuple(img_res), base, image_dtype)
    if cache is not None and key in cache:
        arr, scale, left, top = cache.pop(key)
        cache[key] = (arr, scale, left, top)            # most recently used last
        intrinsics[:2, :] *= scale
        intrinsics[0, -1] -= left
        intrinsics[1, -1] -= top
        return arr, intrinsics
    img = Image.open(path)
    w_src, h_src = img.size
    h_t, w_t = img_res
    sh, sw = float(h_t) / h_src, float(w_t) / w_src
    if sh > 1 or sw > 1:
        raise ValueError("target resolution must not exceed the image resolution")
    scale = max(sh, sw)
    img = img.resize(size=(int(w_src * scale), int(h_src * scale)), resample=Image.BILINEAR)
    w_r, h_r = img.size
    intrinsics[:2, :] *= scale
    final_h = h_t if h_r > h_t else int(math.floor(h_t / base) * base)
    final_w = w_t if w_r > w_t else int(math.floor(w_t / base) * base)
    top = int(math.floor((h_r - final_h) / 2))
    left = int(math.floor((w_r - final_w) / 2))
    img = img.crop((left, top, left + final_w, top + final_h))
    intrinsics[0, -1] -= left
    intrinsics[1, -1] -= top
    arr = np.array(img, dtype=np.float32) / 255.0 if image_dtype == "float32" else np.asarray(img, dtype=np.uint8)
    if arr.ndim == 2:
        arr = np.dstack((arr, arr, arr))
    if cache is not None and cache_size > 0:
        cache[key] = (arr, scale, left, top)
        while len(cache) > cache_size:
            cache.pop(next(iter(cache)))
    return arr, intrinsics


class EvalDataset:
    """Same constructor keywords and item dict as the reference's eval MVSDataset."""

    def __init__(self, datapath, listfile, mode="test", nviews=5, ndepths=192, interval_scale=1.06,
                 pairfile="pair.txt", cam_subfolder="Cameras", img_subfolder="Rectified/{}/rect_{:0>3}_3_r5000.png",
                 img_res=(512, 640), dataset_name="dtu", cache_images: int = 0, image_dtype: str = "float32"):
        assert mode == "test"
        # "uint8": items carry the decoded 8-bit pixels ([N,3,H,W] uint8) instead of the reference's floats; the
        # drop-in MVSNet.forward accepts them (same bits after its on-device division by 255)
        if image_dtype not in ("float32", "uint8"):
            raise ValueError(f"image_dtype must be 'float32' or 'uint8', got {image_dtype!r}")
        self.image_dtype = image_dtype
        # decoded-image LRU (0 = off, the reference's behaviour: every sample decodes its views again)
        self.cache_images = int(cache_images)
        self._img_cache = {} if self.cache_images > 0 else None
        self._cam_cache = {}     # cam file -> parsed (intrinsics, extrinsics, depth_min, interval); used by assemble()
        self.datapath, self.nviews, self.ndepths = datapath, nviews, ndepths
        self.interval_scale, self.cam_subfolder, self.img_subfolder = interval_scale, cam_subfolder, img_subfolder
        self.img_res, self.dataset_name = img_res, dataset_name
        with open(listfile) as f:
            scans = [ln.rstrip() for ln in f.readlines()]
        pair_path = os.path.join(datapath, "../..", pairfile) if dataset_name == "bin" \
            else os.path.join(datapath, pairfile)
        pairs = parse_pair_file(pair_path)
        self.metas = [(scan, ref, src) for scan in scans for ref, src in pairs]

    def __len__(self):
        return len(self.metas)

    # ---- the sample in three steps (used whole by __getitem__, piecewise by decoder_pool.ViewDecoderPool,
    # which decodes every view ONCE and shares it between the samples that use it) ----------------
    def view_plan(self, idx):
        """-> (filename template, [(image path, cam path), ...]) of sample idx: reference view first."""
        scan, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.nviews - 1]
        views = []
        for vid in view_ids:
            img_vid = vid + 1 if self.dataset_name in ("dtu",) else vid
            views.append((os.path.join(self.datapath, self.img_subfolder.format(scan, img_vid)),
                          os.path.join(self.datapath, self.cam_subfolder, "{:0>8}_cam.txt".format(vid))))
        return scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}", views

    def decode_view(self, img_path):
        """The expensive part: decode + rescale + crop one image.  -> (float32 (or uint8) [3,H,W] contiguous in [0,1],
        (scale, left, top)): what the intrinsics of any sample using this view are adjusted by."""
        probe = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]], dtype=np.float64)
        arr, probe = load_image_rescaled_cropped(img_path, probe, img_res=self.img_res, cache=self._img_cache,
                                                 cache_size=self.cache_images, image_dtype=self.image_dtype)
        return np.ascontiguousarray(arr.transpose(2, 0, 1)), (float(probe[0, 0]), float(-probe[0, 2]), float(-probe[1, 2]))

    def assemble(self, idx, adjust):
        """Everything of sample idx except the pixels; adjust[i] = (scale, left, top) of view i."""
        filename, views = self.view_plan(idx)
        projs, intr_list, extr_list = [], [], []
        depth_values = None
        for i, (_, cam_path) in enumerate(views):
            if cam_path not in self._cam_cache:
                self._cam_cache[cam_path] = parse_cam_file(cam_path, self.interval_scale)
            intr, extr, dmin, dint = self._cam_cache[cam_path]
            intr, extr = intr.copy(), extr.copy()
            scale, left, top = adjust[i]
            intr[:2, :] *= scale            # the same in-place float32 updates as load_image_rescaled_cropped
            intr[0, -1] -= left
            intr[1, -1] -= top
            intr[:2, :] /= 4.0
            intr_list.append(intr)
            extr_list.append(extr)
            proj = extr.copy()
            proj[:3, :4] = np.matmul(intr, proj[:3, :4])
            projs.append(proj)
            if i == 0:
                depth_values = np.arange(dmin, dint * (self.ndepths - 0.5) + dmin, dint, dtype=np.float32)
        return {"proj_matrices": np.stack(projs), "intrinsics": intr_list, "extrinsics": extr_list,
                "depth_values": depth_values, "filename": filename}

    def __getitem__(self, idx):
        scan, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.nviews - 1]
        imgs, projs, intr_list, extr_list = [], [], [], []
        depth_values = None
        for i, vid in enumerate(view_ids):
            img_vid = vid + 1 if self.dataset_name in ("dtu",) else vid
            img_path = os.path.join(self.datapath, self.img_subfolder.format(scan, img_vid))
            cam_path = os.path.join(self.datapath, self.cam_subfolder, "{:0>8}_cam.txt".format(vid))
            intr, extr, dmin, dint = parse_cam_file(cam_path, self.interval_scale)
            img, intr = load_image_rescaled_cropped(img_path, intr, img_res=self.img_res, cache=self._img_cache,
                                                    cache_size=self.cache_images, image_dtype=self.image_dtype)
            imgs.append(img)
            intr[:2, :] /= 4.0  # feature scale (the network downsamples by 4)
            intr_list.append(intr)
            extr_list.append(extr)
            proj = extr.copy()
            proj[:3, :4] = np.matmul(intr, proj[:3, :4])
            projs.append(proj)
            if i == 0:
                depth_values = np.arange(dmin, dint * (self.ndepths - 0.5) + dmin, dint, dtype=np.float32)
        return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]),
                "proj_matrices": np.stack(projs),
                "intrinsics": intr_list,
                "extrinsics": extr_list,
                "depth_values": depth_values,
                "filename": scan + "/{}/" + "{:0>8}".format(view_ids[0]) + "{}"}
