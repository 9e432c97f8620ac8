#!/usr/bin/env python3
"""The drop-in MVSNet.forward from images (FeatureNet in HIP + path), cfg2, in a loop -- for rocprofv3 runs."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from scene_3dreconstruction_mvsnet_amd import MVSNet, synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = synthetic.CONFIGS["cfg2"]
N, H, W, D = cfg["nviews"], cfg["H"], cfg["W"], cfg["D"]
dev = torch.device("cuda:0")
model = MVSNet(refine=False)
synthetic.randomize_bn_(model, seed=0)
model = model.to(dev).eval()
imgs, proj, dv = synthetic.make_inputs(N, H, W, D, seed=0, interval_scale=cfg["interval_scale"])
imgs_d, proj_d, dv_d = (torch.from_numpy(a).to(dev) for a in (imgs, proj, dv))
for _ in range(reps):
    model(imgs_d, proj_d, dv_d)
torch.cuda.synchronize()
