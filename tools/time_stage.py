#!/usr/bin/env python3
"""Time one stage with HIP events: python3 tools/time_stage.py conv0 [reps]  (diagnostics)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from scene_3dreconstruction_mvsnet_amd import _lib, synthetic  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "conv0"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = synthetic.CONFIGS["cfg2"]
N, D, h, w = cfg["nviews"], cfg["D"], cfg["H"] // 4, cfg["W"] // 4
dev = torch.device("cuda:0")
feats = torch.from_numpy(synthetic.random_features(N, 32, h, w, seed=0)).to(dev)
proj = torch.from_numpy(synthetic.cameras(N, h, w)).to(dev)
dv = torch.from_numpy(synthetic.depth_values(D)).to(dev)
blob = _lib.pack_weights(synthetic.random_costreg_state(0)).to(dev)
ws = _lib.alloc_workspace(N, 32, D, h, w, dev)
rt = _lib.relative_proj(proj)
if what == "warpz":   # all-zero features: same instruction stream, idle data paths (DVFS / power check)
    feats = torch.zeros_like(feats)
var = _lib.warp_variance(feats, rt, dv, ws)
layer = {"conv0": 0, "conv0z": 0, "conv1": 1, "conv2": 2, "prob": 10}.get(what)
if what == "conv0z":   # all-zero input: same instruction stream, idle data paths (DVFS / power check)
    var = torch.zeros_like(var)


def run():
    if what == "tail":
        _lib.conv11_prob(t8, t0, blob)
    elif what in ("warp", "warpz"):
        _lib.warp_variance(feats, rt, dv, ws)
    else:
        _lib.conv_layer(layer, var if layer == 0 else x_in, None, blob)


x_in = None
if what == "tail":   # conv11 + skip + prob on random inputs of the cfg2 shapes
    t8 = torch.randn((2, D // 2, h // 2, w // 2, 8), device=dev)
    t0 = torch.randn((1, D, h, w, 8), device=dev)
if layer == 10:
    x_in = torch.randn((1, D, h, w, 8), device=dev)
elif layer in (1, 2):
    x_in = _lib.conv_layer(0, var, None, blob)
    if layer == 2:
        x_in = _lib.conv_layer(1, x_in, None, blob)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
print(f"{what} lib={os.path.basename(_lib.LIB_PATH)} {e0.elapsed_time(e1) / reps:.4f} ms")
