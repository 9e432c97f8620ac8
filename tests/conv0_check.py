#!/usr/bin/env python3
"""Child-process helper of tests/test_gpu_fullsize.py::test_cfg2_conv0_kernel_variants_match_oracle: conv0 at the bench
size (cfg2) against the oracle with the bounds of test_cfg2_every_layer_matches_oracle[0], and on the heavy-tailed volume
of test_winograd_layers_on_a_heavy_tailed_nonnegative_volume[0], under whatever MVS_CONV0_SPLIT the parent set (the
kernel selection is read once per process)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from conftest import rel_l1  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from scene_3dreconstruction_mvsnet_amd import _lib, synthetic  # noqa: E402

DEV = "cuda:0"
LAYER_ATOL = 2e-4


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def main():
    c = synthetic.CONFIGS["cfg2"]
    N, h, w, D = c["nviews"], c["H"] // 4, c["W"] // 4, c["D"]
    sd = synthetic.random_costreg_state(seed=0)
    blob = _lib.pack_weights(sd).to(DEV)
    var = orc.variance_volume(synthetic.random_features(N, 32, h, w, seed=0), synthetic.cameras(N, h, w),
                              synthetic.depth_values(D, interval_scale=c["interval_scale"]))
    want = orc.conv3d(var, sd["conv0.conv.weight"], bn=orc._bn(sd, "conv0.bn"))
    got = _lib.from_c8(_lib.conv_layer(0, _lib.to_c8(cu(var)), None, blob)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=LAYER_ATOL * max(float(np.abs(want).max()), 1.0))
    r = rel_l1(got, want)
    assert r < 2e-6, r
    # heavy-tailed non-negative volume: |err| <= 2.5e-7 x local input scale x weight mass
    g = np.random.default_rng(77)
    x = np.exp(3.0 * g.standard_normal((32, 48, 64, 96))).astype(np.float32)
    want = orc.conv3d(x, sd["conv0.conv.weight"], bn=orc._bn(sd, "conv0.bn"))
    got = _lib.from_c8(_lib.conv_layer(0, _lib.to_c8(cu(x)), None, blob)).cpu().numpy()
    m = torch.from_numpy(x.max(axis=0))[None, None]
    m = torch.nn.functional.max_pool3d(m, kernel_size=(9, 3, 3), stride=1, padding=(4, 1, 1))[0, 0].numpy()
    gamma, beta, mean, var_ = orc._bn(sd, "conv0.bn")
    wmass = np.abs(sd["conv0.conv.weight"]).reshape(want.shape[0], -1).sum(1) * np.abs(gamma) / np.sqrt(var_ + 1e-5)
    worst = float((np.abs(got - want) / (2.5e-7 * wmass[:, None, None, None] * m[None] + 1e-6)).max())
    print(f"conv0_check MVS_CONV0_SPLIT={os.environ.get('MVS_CONV0_SPLIT')} cfg2 rel-L1 {r:.2e}; heavy-tailed max err / bound {worst:.3f}")
    return 0 if worst <= 1.0 else 1


if __name__ == "__main__":
    sys.exit(main())
