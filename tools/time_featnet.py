"""Per-layer and whole-net timing of FeatureNet in HIP (featnet.hip) at a config's image size, with
PyTorch-ROCm (MIOpen) timed beside it.  Usage: python tools/time_featnet.py [cfg2]"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from scene_3dreconstruction_mvsnet_amd import MVSNet, _lib, synthetic  # noqa: E402

cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
N, H, W = cfg["nviews"], cfg["H"], cfg["W"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = MVSNet(refine=False)
synthetic.randomize_bn_(model, seed=0)
model = model.to(dev).eval()
fblob = model._feature_blob(dev)
imgs = torch.rand(N, 3, H, W, device=dev)


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


res = {}
x = imgs
hh, ww = H, W
flops_total = 0
for l, (ci, co, k, s) in enumerate(_lib.FEATURE_LAYERS):
    ms = timed(lambda: _lib.feature_layer(l, x, fblob))
    y = _lib.feature_layer(l, x, fblob)
    ho, wo = y.shape[2], y.shape[3]
    fl = 2.0 * N * ho * wo * ci * co * k * k
    by = 4.0 * N * (hh * ww * (3 if l == 0 else ci) + ho * wo * co)
    flops_total += fl
    res[f"L{l}"] = {"ms": round(ms, 4), "TF": round(fl / ms / 1e9, 1), "GBps": round(by / ms / 1e6, 0)}
    x, hh, ww = y, ho, wo
ws = torch.empty(_lib.query_feature_workspace(N, H, W), dtype=torch.uint8, device=dev)
res["hip_total_ms"] = round(timed(lambda: _lib.feature_net(imgs, fblob, ws)), 4)
with torch.no_grad():
    res["torch_total_ms"] = round(timed(lambda: model.feature(imgs)), 4)
res["GFLOP"] = round(flops_total / 1e9, 2)
print(json.dumps(res))
