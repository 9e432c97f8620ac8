"""Seeded synthetic inputs for the MVSNet depth-inference path.

No dataset or checkpoint ships with the reference, so tests, fixtures and bench.py
all draw from the same recipe:

* images   : bilinear-upsampled low-res noise (smooth textures) in [0, 1]
* cameras  : DTU-like pinhole rig, feature-scale intrinsics, projection matrices built
             exactly as datasets/dataloader_eval.py:158-159 (`proj[:3,:4] = K @ E[:3,:4]`)
* depths   : `np.arange(dmin, dint*(D-0.5)+dmin, dint)` as datasets/dataloader_eval.py:163
* weights  : default torch init + randomised BatchNorm statistics and a gain on the
             last conv so the soft-argmin is not flat (default init alone gives a
             degenerate uniform softmax)

Everything here is numpy / torch-CPU and deterministic under the seed.
"""
from __future__ import annotations

import numpy as np
import torch

# (N views, image H, image W, D) for BASELINE.json configs
CONFIGS = {
    "cfg1": dict(nviews=3, H=128, W=160, D=48, interval_scale=1.06),
    "cfg2": dict(nviews=5, H=512, W=640, D=192, interval_scale=1.06),
    "cfg3": dict(nviews=5, H=1184, W=1600, D=256, interval_scale=1.06),
    "cfg5": dict(nviews=4, H=512, W=640, D=192, interval_scale=1.33),
}


def depth_values(D: int, dmin: float = 425.0, interval: float = 2.5,
                 interval_scale: float = 1.06) -> np.ndarray:
    """float32 depth hypotheses, formula of datasets/dataloader_eval.py:163."""
    dint = np.float32(interval * interval_scale)
    dv = np.arange(dmin, dint * (D - 0.5) + dmin, dint, dtype=np.float32)
    assert dv.shape[0] == D, (dv.shape, D)
    return dv


def cameras(nviews: int, h: int, w: int, baseline=(-30.0, 5.0, 0.0),
            yaw_deg: float = 0.0) -> np.ndarray:
    """[N,4,4] float32 projection matrices at feature scale (h x w).

    K = [[361.5*w/160, 0, w/2], [0, 360*h/128, h/2], [0,0,1]],
    E_i = [R_i | t_i] with t_i = i * baseline (mm) and an optional yaw of i*yaw_deg.
    """
    K = np.array([[361.5 * w / 160.0, 0.0, w / 2.0],
                  [0.0, 360.0 * h / 128.0, h / 2.0],
                  [0.0, 0.0, 1.0]], dtype=np.float32)
    projs = []
    for i in range(nviews):
        E = np.eye(4, dtype=np.float32)
        a = np.deg2rad(yaw_deg * i)
        R = np.array([[np.cos(a), 0.0, np.sin(a)],
                      [0.0, 1.0, 0.0],
                      [-np.sin(a), 0.0, np.cos(a)]], dtype=np.float32)
        E[:3, :3] = R
        E[:3, 3] = np.asarray(baseline, dtype=np.float32) * i
        P = E.copy()
        P[:3, :4] = K @ E[:3, :4]
        projs.append(P)
    return np.stack(projs).astype(np.float32)


def in_image_fraction(proj: np.ndarray, dv: np.ndarray, h: int, w: int) -> float:
    """Fraction of the (pixel, depth, source view) sampling points of the homography warp
    (models/module.py:107-136) that fall inside the source image, i.e. have at least one in-bounds
    bilinear tap.  Host-side bookkeeping in float64, not part of the data path."""
    proj = np.asarray(proj, np.float64)
    n = proj.shape[0]
    if n < 2:
        return 1.0
    dv = np.asarray(dv, np.float64)
    y, x = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    xyz = np.stack([x.ravel(), y.ravel(), np.ones(h * w)])                 # [3, hw]
    inside = 0
    for v in range(1, n):
        rel = proj[v] @ np.linalg.inv(proj[0])
        q = rel[:3, :3] @ xyz                                               # [3, hw]
        p = q[:, None, :] * dv[None, :, None] + rel[:3, 3][:, None, None]   # [3, D, hw]
        with np.errstate(all="ignore"):
            ix = p[0] / p[2] * (w / (w - 1.0)) - 0.5
            iy = p[1] / p[2] * (h / (h - 1.0)) - 0.5
        inside += int(((ix > -1) & (ix < w) & (iy > -1) & (iy < h)).sum())
    return inside / float((n - 1) * dv.shape[0] * h * w)


def smooth_images(nviews: int, H: int, W: int, seed: int = 0, coarse: int = 8) -> np.ndarray:
    """[N,3,H,W] float32 in [0,1]: seeded low-res noise, bilinear upsampled."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand(nviews, 3, max(2, H // coarse), max(2, W // coarse), generator=g)
    hi = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False)
    return hi.clamp_(0, 1).numpy().astype(np.float32)


def random_features(nviews: int, C: int, h: int, w: int, seed: int = 0,
                    coarse: int = 2) -> np.ndarray:
    """[N,C,h,w] float32 smooth-ish N(0,1) feature maps for path-only runs."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.randn(nviews, C, max(2, h // coarse), max(2, w // coarse), generator=g)
    hi = torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)
    return hi.numpy().astype(np.float32)


def randomize_bn_(model: torch.nn.Module, seed: int = 0, prob_gain: float = 30.0) -> None:
    """In-place: seeded BN statistics/affine and a gain on cost_regularization.prob.weight."""
    g = torch.Generator().manual_seed(1000 + seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                n = m.num_features
                m.weight.copy_(torch.rand(n, generator=g) + 0.5)
                m.bias.copy_(torch.randn(n, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(n, generator=g) + 0.5)
        model.cost_regularization.prob.weight.mul_(prob_gain)


def random_costreg_state(seed: int = 0, prob_gain: float = 30.0) -> dict:
    """CostRegNet-shaped state dict (numpy) without constructing any nn.Module.

    Used by bench.py / large-size tests where only shapes and scale matter.
    Keys follow the reference checkpoint naming (models/mvsnet.py:33-62).
    """
    g = torch.Generator().manual_seed(2000 + seed)
    sd = {}

    def conv_w(name, cout, cin, transposed=False):
        fan_in = cin * 27
        bound = 1.0 / np.sqrt(fan_in)
        shape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
        sd[name] = ((torch.rand(shape, generator=g) * 2 - 1) * bound).numpy()

    def bn(prefix, n):
        sd[prefix + ".weight"] = (torch.rand(n, generator=g) + 0.5).numpy()
        sd[prefix + ".bias"] = (torch.randn(n, generator=g) * 0.1).numpy()
        sd[prefix + ".running_mean"] = (torch.randn(n, generator=g) * 0.1).numpy()
        sd[prefix + ".running_var"] = (torch.rand(n, generator=g) + 0.5).numpy()

    chans = [(32, 8), (8, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)]
    for i, (ci, co) in enumerate(chans):
        conv_w(f"conv{i}.conv.weight", co, ci)
        bn(f"conv{i}.bn", co)
    for name, ci, co in (("conv7", 64, 32), ("conv9", 32, 16), ("conv11", 16, 8)):
        conv_w(f"{name}.0.weight", co, ci, transposed=True)
        bn(f"{name}.1", co)
    conv_w("prob.weight", 1, 8)
    sd["prob.weight"] = sd["prob.weight"] * prob_gain
    sd["prob.bias"] = ((torch.rand(1, generator=g) * 2 - 1) / np.sqrt(8 * 27)).numpy()
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in sd.items()}


def make_inputs(nviews: int, H: int, W: int, D: int, seed: int = 0, batch: int = 1,
                interval_scale: float = 1.06, baseline=(-30.0, 5.0, 0.0), yaw_deg: float = 0.0):
    """(imgs [B,N,3,H,W], proj_matrices [B,N,4,4], depth_values [B,D]) float32 numpy."""
    imgs = np.stack([smooth_images(nviews, H, W, seed=seed + 17 * b) for b in range(batch)])
    proj = np.stack([cameras(nviews, H // 4, W // 4, baseline=baseline, yaw_deg=yaw_deg)
                     for _ in range(batch)])
    dv = np.stack([depth_values(D, interval_scale=interval_scale) for _ in range(batch)])
    return imgs, proj, dv
