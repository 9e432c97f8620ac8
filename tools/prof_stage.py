#!/usr/bin/env python3
"""Run one stage of the path a few times (for rocprofv3 --pmc / --kernel-trace runs).

    python3 tools/prof_stage.py warp|conv0|all [reps] [cfg2|cfg3|cfg5] [f32|f16|bf16]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from scene_3dreconstruction_mvsnet_amd import _lib, synthetic  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg_name = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
storage = sys.argv[4] if len(sys.argv) > 4 else {"cfg3": "bf16", "cfg5": "f16"}.get(cfg_name, "f32")
dt = _lib.dtype_code(storage)
cfg = synthetic.CONFIGS[cfg_name]
N, D, h, w = cfg["nviews"], cfg["D"], cfg["H"] // 4, cfg["W"] // 4
dev = torch.device("cuda:0")
feats = torch.from_numpy(synthetic.random_features(N, 32, h, w, seed=0)).to(dev)
proj = torch.from_numpy(synthetic.cameras(N, h, w)).to(dev)
dv = torch.from_numpy(synthetic.depth_values(D, interval_scale=cfg["interval_scale"])).to(dev)
blob = _lib.pack_weights(synthetic.random_costreg_state(0)).to(dev)
ws = _lib.alloc_workspace(N, 32, D, h, w, dev, dt)
rt = _lib.relative_proj(proj)
var = _lib.warp_variance(feats, rt, dv, ws, dtype=dt)
depth = torch.empty((h, w), device=dev)
conf = torch.empty_like(depth)
torch.cuda.synchronize()
for _ in range(reps):
    if what == "warp":
        var = _lib.warp_variance(feats, rt, dv, ws, dtype=dt)
    elif what == "conv0":
        _lib.conv_layer(0, var, None, blob, dtype=dt)
    else:
        _lib.depth_infer(feats, proj, dv, blob, ws, depth, conf, dtype=dt)
torch.cuda.synchronize()
