#!/usr/bin/env python3
"""Helper for tests/test_gpu_parity.py::test_optin_kernel_variants (run in a child process so
that the kernel-selection environment variables, read once per process, take effect).
Checks the whole HIP path against the reference fixtures and one ragged random problem."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from conftest import load_fixture, load_weights, rel_l1  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from scene_3dreconstruction_mvsnet_amd import _lib, synthetic  # noqa: E402

DEV = "cuda:0"


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def infer(feats, proj, dv, sd):
    N, C, h, w = feats.shape
    D = dv.shape[0]
    ws = _lib.alloc_workspace(N, C, D, h, w, DEV)
    depth = torch.empty((h, w), dtype=torch.float32, device=DEV)
    conf = torch.empty_like(depth)
    _lib.depth_infer(cu(feats), cu(proj), cu(dv), _lib.pack_weights(sd).to(DEV), ws, depth, conf)
    torch.cuda.synchronize()
    return depth.cpu().numpy(), conf.cpu().numpy()


def main():
    sd = orc.costreg_state(load_weights())
    worst = 0.0
    for name in ("small", "n5yaw", "oob", "cfg1"):
        fx = load_fixture(name)
        depth, _ = infer(fx["features"][0], fx["proj_matrices"][0], fx["depth_values"][0], sd)
        worst = max(worst, rel_l1(depth, fx["depth"][0]))
    # ragged tiles (w = 40 is not a multiple of 16/32) and a rotated rig, against the oracle
    feats = synthetic.random_features(4, 32, 24, 40, seed=3)
    proj = synthetic.cameras(4, 24, 40, yaw_deg=2.0)
    dv = synthetic.depth_values(16)
    sd2 = synthetic.random_costreg_state(seed=6)
    depth, _ = infer(feats, proj, dv, sd2)
    depth_o, _ = orc.depth_infer(feats, proj, dv, sd2)
    worst = max(worst, rel_l1(depth, depth_o))
    print(f"variant env={ {k: v for k, v in os.environ.items() if k.startswith('MVS_')} } worst rel-L1 {worst:.3e}")
    return 0 if worst < 1e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
