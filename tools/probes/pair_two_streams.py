#!/usr/bin/env python3
"""Two stages on two HIP streams from two host threads -- thread A: warp + variance (tap-cache kernel), thread B: the
fused tail -- every iteration's output compared bit for bit with the single-stream result.
python3 tools/probes/pair_two_streams.py [D h w] [reps]"""
import os
import sys
import threading

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from scene_3dreconstruction_mvsnet_amd import _lib, synthetic  # noqa: E402

D, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (48, 32, 40)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
N = 3
dev = torch.device("cuda:0")
blob = _lib.pack_weights(synthetic.random_costreg_state(0)).to(dev)
feats = torch.from_numpy(synthetic.random_features(N, 32, h, w, seed=1)).to(dev)
proj = torch.from_numpy(synthetic.cameras(N, h, w)).to(dev)
dv = torch.from_numpy(synthetic.depth_values(D)).to(dev)
rt = _lib.relative_proj(proj)
g = torch.Generator().manual_seed(5)
t8 = torch.rand((2, D // 2, h // 2, w // 2, 8), generator=g).to(dev)
t0 = torch.rand((1, D, h, w, 8), generator=g).to(dev)
wsA = _lib.alloc_workspace(N, 32, D, h, w, dev)
want_var = _lib.warp_variance(feats, rt, dv, wsA).clone()
want_cost = _lib.conv11_prob(t8, t0, blob).clone()
torch.cuda.synchronize()
for rep in range(reps):
    bad = {"warp": 0, "tail": 0}

    def wa():
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            ws = _lib.alloc_workspace(N, 32, D, h, w, dev)
            outs = [_lib.warp_variance(feats, rt, dv, ws).clone() for _ in range(150)]
        st.synchronize()
        bad["warp"] = sum(not torch.equal(o, want_var) for o in outs)
        for j, o in enumerate(outs):
            if not torch.equal(o, want_var):
                neq = (o != want_var) | (o.isnan() != want_var.isnan())
                idx = neq.nonzero()
                print("  warp iteration", j, "differs in", int(neq.sum()), "elements; chunks", idx[:, 0].unique().tolist(),
                      "depths", idx[:, 1].unique().tolist()[:12], "rows", idx[:, 2].unique().tolist()[:12],
                      "cols", idx[:, 3].min().item(), "-", idx[:, 3].max().item(), "ch", idx[:, 4].unique().tolist())
                k = tuple(idx[0].tolist())
                print("   first:", k, "got", o[k].item(), "want", want_var[k].item(), " got is a value of want elsewhere:",
                      bool((want_var == o[k]).any()))
                break

    def wb():
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            outs = [_lib.conv11_prob(t8, t0, blob).clone() for _ in range(150)]
        st.synchronize()
        bad["tail"] = sum(not torch.equal(o, want_cost) for o in outs)

    ths = [threading.Thread(target=f) for f in (wa, wb)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    print(f"{D}x{h}x{w} rep {rep}: mismatching iterations of 60:", bad)
