#!/usr/bin/env python3
"""Run FeatureNet (HIP) a few times on cfg2-sized images (for rocprofv3 --kernel-trace --stats).
    python3 tools/prof_featnet.py [reps] [u8]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from scene_3dreconstruction_mvsnet_amd import _lib, synthetic  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = synthetic.CONFIGS["cfg2"]
N, H, W = cfg["nviews"], cfg["H"], cfg["W"]
dev = torch.device("cuda:0")
imgs = torch.rand((N, 3, H, W), device=dev)
if len(sys.argv) > 2 and sys.argv[2] == "u8":
    imgs = (imgs * 255).to(torch.uint8)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_weights  # noqa: E402  (the committed reference-initialised weights, tests/golden)

w = load_weights()
fblob = _lib.pack_feature_weights({k[len("feature."):]: v for k, v in w.items() if k.startswith("feature.")}).to(dev)
for _ in range(3):
    _lib.feature_net(imgs, fblob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    _lib.feature_net(imgs, fblob)
e1.record()
torch.cuda.synchronize()
print(f"feature_net N={N} {H}x{W}: {e0.elapsed_time(e1) / reps:.4f} ms")
