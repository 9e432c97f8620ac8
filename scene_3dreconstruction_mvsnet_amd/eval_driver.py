"""Sharded depth-generation driver: the MI355X counterpart of the reference's `save_depth`
(eval.py:283-500) for the files the downstream `filter_depth` stage reads.

Per sample it runs the drop-in MVSNet forward and writes, under `outdir`:
    {scan}/depth_est/{view:08d}.pfm     <- outputs["depth"][b]                  (eval.py:387)
    {scan}/confidence/{view:08d}.pfm    <- outputs["photometric_confidence"][b] (eval.py:392)
    {scan}/cams/{view:08d}_cam.txt      <- write_cam(K, E, ["000","2.5","",""]) (eval.py:396,107-126)
    {scan}/images/{view:08d}.png        <- uint8(ref image * 255), RGB          (eval.py:346-350)
with `filename = "{scan}/{{}}/{view:08d}{{}}"` as the reference datasets produce it
(datasets/dataloader_eval.py:176).  Not reproduced (flagged, not silently changed): the PNG previews
(eval.py:388,393, need cv2), the point-cloud accumulation (eval.py:409-440, needs open3d).

Sharding replaces `nn.DataParallel` (eval.py:309): rank r of R processes dataset items r::R
(scene_3dreconstruction_mvsnet_amd.sharding); every rank writes its own files, so no collective is
needed here.  File writes happen on a writer thread so that they overlap the next forward pass.
"""
from __future__ import annotations

import os
import queue
import threading

import numpy as np
import torch
from PIL import Image

from . import data_io, sharding


def write_cam(file: str, K, R, depth_params) -> None:
    """Same text as the reference's write_cam (eval.py:107-126)."""
    with open(file, "w") as f:
        f.write("extrinsic\n")
        for i in range(4):
            for j in range(4):
                f.write(str(R[i][j]) + " ")
            f.write("\n")
        f.write("\n")
        f.write("intrinsic\n")
        for i in range(3):
            for j in range(3):
                f.write(str(K[i][j]) + " ")
            f.write("\n")
        f.write("\n" + str(depth_params[0]) + " " + str(depth_params[1]) + " " + str(depth_params[2]) +
                " " + str(depth_params[3]) + "\n")


def _writer(q: "queue.Queue"):
    while True:
        job = q.get()
        if job is None:
            return
        outdir, filename, depth, conf, K, E, img = job
        depth_fn, conf_fn = data_io.depth_map_paths(outdir, filename)
        cam_fn = os.path.join(outdir, filename.format("cams", "_cam.txt"))
        img_fn = os.path.join(outdir, filename.format("images", ".png"))
        for fn in (depth_fn, conf_fn, cam_fn, img_fn):
            os.makedirs(os.path.dirname(fn), exist_ok=True)
        if img is not None:
            # eval.py:346-350: the two BGR<->RGB swaps there cancel, the file holds uint8(img*255) RGB
            Image.fromarray(np.uint8(np.transpose(img, (1, 2, 0)) * 255)).save(img_fn)
        data_io.save_pfm(depth_fn, depth)
        data_io.save_pfm(conf_fn, conf)
        if K is not None and E is not None:
            write_cam(cam_fn, K=K, R=E, depth_params=["000", "2.5", "", ""])


def save_depth_sharded(model, dataset, outdir: str, rank: int = 0, world: int = 1, device=None):
    """Run `model` over dataset items rank::world and write the reference's per-view files.

    dataset[i] -> dict with "imgs" [N,3,H,W], "proj_matrices" [N,4,4], "depth_values" [D],
    "filename" and optionally "intrinsics" / "extrinsics" (lists; entry 0 = reference view), as
    the reference's eval datasets return them (datasets/dataloader_eval.py:171-176).
    Returns the list of dataset indices this rank processed.
    """
    device = device or torch.device("cuda", torch.cuda.current_device())
    model = model.to(device).eval()
    mine = sharding.shard_units(len(dataset), rank, world)
    q: "queue.Queue" = queue.Queue(maxsize=8)
    th = threading.Thread(target=_writer, args=(q,), daemon=True)
    th.start()
    try:
        with torch.no_grad():
            for idx in mine:
                s = dataset[idx]
                imgs = torch.as_tensor(np.asarray(s["imgs"]), dtype=torch.float32)[None].to(device)
                proj = torch.as_tensor(np.asarray(s["proj_matrices"]), dtype=torch.float32)[None].to(device)
                dv = torch.as_tensor(np.asarray(s["depth_values"]), dtype=torch.float32)[None].to(device)
                out = model(imgs, proj, dv)
                depth = out["depth"][0].detach().cpu().numpy().copy()          # utils.py:55
                conf = out["photometric_confidence"][0].detach().cpu().numpy().copy()
                K = np.asarray(s["intrinsics"][0]) if "intrinsics" in s else None
                E = np.asarray(s["extrinsics"][0]) if "extrinsics" in s else None
                q.put((outdir, s["filename"], depth, conf, K, E, np.asarray(s["imgs"][0], np.float32)))
    finally:
        q.put(None)
        th.join()
    return mine
