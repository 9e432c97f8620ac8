#!/usr/bin/env python3
"""Two host threads on two HIP streams drive one MVSNet module (tests/test_gpu_parity.py::
test_two_host_threads_on_two_streams_share_one_module): count the iterations whose depth map is not bit-identical to the
single-stream result, under whatever kernel-selection environment the caller set."""
import os
import sys
import threading

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_weights  # noqa: E402
from scene_3dreconstruction_mvsnet_amd import MVSNet, synthetic  # noqa: E402

DEV = "cuda:0"
w = load_weights()
model = MVSNet(refine=False).to(DEV).eval()
model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
problems = []
for seed in (1, 2):
    imgs, proj, dv = synthetic.make_inputs(3, 128, 160, 48, seed=seed)
    problems.append(tuple(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV) for a in (imgs, proj, dv)))
want = [model(*p)["depth"].clone() for p in problems]
again = [model(*p)["depth"].clone() for p in problems]
torch.cuda.synchronize()
print("sequential repeat identical:", [bool(torch.equal(a, b)) for a, b in zip(want, again)])
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    got = [[], []]

    def worker(i):
        st = torch.cuda.Stream(DEV)
        with torch.cuda.stream(st):
            for _ in range(20):
                got[i].append(model(*problems[i])["depth"])
        st.synchronize()

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    bad = [(i, j, float((d - want[i]).abs().max()), int((d != want[i]).sum())) for i in range(2) for j, d in enumerate(got[i])
           if not torch.equal(d, want[i])]
    print("rep", rep, "mismatching iterations:", len(bad), bad[:4])
