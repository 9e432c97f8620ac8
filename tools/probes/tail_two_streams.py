#!/usr/bin/env python3
"""The fused tail (mvs_conv11_prob) alone on two HIP streams from two host threads: iterations whose logits are not
bit-identical to the single-stream result.  python3 tools/probes/tail_two_streams.py [D h w] [reps]"""
import os
import sys
import threading

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from scene_3dreconstruction_mvsnet_amd import _lib, synthetic  # noqa: E402

D, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (48, 32, 40)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
blob = _lib.pack_weights(synthetic.random_costreg_state(0)).to(dev)
g = torch.Generator().manual_seed(5)
ins = [(torch.rand((2, D // 2, h // 2, w // 2, 8), generator=g).to(dev), torch.rand((1, D, h, w, 8), generator=g).to(dev))
       for _ in range(2)]
want = [_lib.conv11_prob(a, b, blob).clone() for a, b in ins]
torch.cuda.synchronize()
for rep in range(reps):
    got = [[], []]

    def worker(i):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            for _ in range(50):
                got[i].append(_lib.conv11_prob(ins[i][0], ins[i][1], blob).clone())
        st.synchronize()

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    bad = [(i, j, float((d - want[i]).abs().max()), int((d != want[i]).sum())) for i in range(2) for j, d in enumerate(got[i])
           if not torch.equal(d, want[i])]
    print(f"{D}x{h}x{w} rep {rep}: mismatching iterations {len(bad)} of 100", bad[:3])
