#!/usr/bin/env python3
"""End-to-end timing of the drop-in MVSNet.forward from images (FeatureNet in HIP or on PyTorch-ROCm,
with / without H2D of the images, + the HIP depth path); for DESIGN.md -- bench.py's `value` is the path-only figure."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from scene_3dreconstruction_mvsnet_amd import MVSNet, synthetic  # noqa: E402

cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
N, H, W, D = cfg["nviews"], cfg["H"], cfg["W"], cfg["D"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = MVSNet(refine=False)
synthetic.randomize_bn_(model, seed=0)
model = model.to(dev).eval()
imgs, proj, dv = synthetic.make_inputs(N, H, W, D, seed=0, interval_scale=cfg["interval_scale"])
imgs_h = torch.from_numpy(imgs).pin_memory()
proj_d, dv_d = torch.from_numpy(proj).to(dev), torch.from_numpy(dv).to(dev)
for impl, mode in (("hip", "resident"), ("hip", "h2d"), ("torch", "resident"), ("torch", "h2d")):
    model.feature_impl = impl
    for _ in range(3):
        model(imgs_h.to(dev), proj_d, dv_d)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    imgs_d = imgs_h.to(dev)
    for _ in range(K):
        if mode == "h2d":
            imgs_d = imgs_h.to(dev, non_blocking=True)
        out = model(imgs_d, proj_d, dv_d)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"FeatureNet={impl} {mode}: {1 / dt:.1f} maps/s ({dt * 1e3:.3f} ms per forward incl. FeatureNet on {N} views)")
