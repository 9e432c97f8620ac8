"""Full-size (BASELINE.json configs) per-stage parity of the HIP path against the CPU oracle.

The end-to-end depth metric is insensitive (zeroing one of conv4's 27 taps moves the cfg2 depth
rel-L1 by 1.4e-5), and some kernels / launch orders are only selected at full size (the persistent
conv1 kernel, the depth-slab-fastest warp order).  So every stage is compared on its own at the
size bench.py measures: the variance volume, each CostRegNet layer fed with the ORACLE's input
(`mvs_conv_layer`), the cost volume and the final maps; reference lines models/mvsnet.py:64-73,
145-177.  All through the C ABI; the oracle needs a few seconds per map on the GPU box's host.
"""
import numpy as np
import pytest
import torch

from conftest import assert_conf_close, load_fixture, rel_l1
from oracle import oracle as orc
from scene_3dreconstruction_mvsnet_amd import _lib, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

LAYER_ATOL = 2e-4       # x max|want|: fp32 summation-order / Winograd re-association noise
STRIDES = {1: 2, 3: 2, 5: 2}
SKIP_OF = {7: "c4", 8: "c2", 9: "c0"}   # layer index -> skip tensor (conv7 / conv9 / conv11)


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def oracle_chain(feats, proj, dv, sd):
    """Every intermediate of the path in the reference's layout (models/mvsnet.py:64-73)."""
    o = {"var": orc.variance_volume(feats, proj, dv)}
    x = o["var"]
    for i in range(7):
        x = orc.conv3d(x, sd[f"conv{i}.conv.weight"], bn=orc._bn(sd, f"conv{i}.bn"), stride=STRIDES.get(i, 1))
        o[f"c{i}"] = x
    o["d7"] = o["c4"] + orc.deconv3d(o["c6"], sd["conv7.0.weight"], bn=orc._bn(sd, "conv7.1"))
    o["d9"] = o["c2"] + orc.deconv3d(o["d7"], sd["conv9.0.weight"], bn=orc._bn(sd, "conv9.1"))
    o["d11"] = o["c0"] + orc.deconv3d(o["d9"], sd["conv11.0.weight"], bn=orc._bn(sd, "conv11.1"))
    o["cost"] = orc.conv3d(o["d11"], sd["prob.weight"], bias=sd["prob.bias"], bn=None, relu=False)[0]
    o["depth"], o["conf"], o["idx"] = orc.softargmin_conf(o["cost"], dv)
    return o


# (input name, output name) of mvs_conv_layer(layer) in the chain above
LAYER_IO = [("var", "c0"), ("c0", "c1"), ("c1", "c2"), ("c2", "c3"), ("c3", "c4"), ("c4", "c5"), ("c5", "c6"),
            ("c6", "d7"), ("d7", "d9"), ("d9", "d11"), ("d11", "cost")]


@pytest.fixture(scope="module")
def cfg2():
    c = synthetic.CONFIGS["cfg2"]
    N, h, w, D = c["nviews"], c["H"] // 4, c["W"] // 4, c["D"]
    feats = synthetic.random_features(N, 32, h, w, seed=0)
    proj = synthetic.cameras(N, h, w)
    dv = synthetic.depth_values(D, interval_scale=c["interval_scale"])
    sd = synthetic.random_costreg_state(seed=0)
    return dict(feats=feats, proj=proj, dv=dv, sd=sd, o=oracle_chain(feats, proj, dv, sd),
                blob=_lib.pack_weights(sd).to(DEV))


def hip_variance_c8(feats, proj, dv, dtype=_lib.MVS_F32):
    N, C, h, w = feats.shape
    ws = _lib.alloc_workspace(N, C, dv.shape[0], h, w, DEV, dtype)
    return _lib.warp_variance(cu(feats), _lib.relative_proj(cu(proj)), cu(dv), ws, dtype=dtype)


def test_cfg2_variance_volume_matches_oracle(cfg2):
    got = _lib.from_c8(hip_variance_c8(cfg2["feats"], cfg2["proj"], cfg2["dv"])).cpu().numpy()
    want = cfg2["o"]["var"]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-4)
    assert rel_l1(got, want) < 1e-5      # measured 2.8e-6 (one v_rcp_f32 in the projection, 1 ulp)


@pytest.mark.parametrize("layer", list(range(11)))
def test_cfg2_every_layer_matches_oracle(cfg2, layer):
    """mvs_conv_layer at the bench size, fed with the oracle's input of that layer: this is where
    the persistent conv1 kernel and the full-size tile schedules of every kernel are compared."""
    o = cfg2["o"]
    src, dst = LAYER_IO[layer]
    x = _lib.to_c8(cu(o[src]))
    skip = _lib.to_c8(cu(o[SKIP_OF[layer]])) if layer in SKIP_OF else None
    y = _lib.conv_layer(layer, x, skip, cfg2["blob"])
    got = (y if layer == 10 else _lib.from_c8(y)).cpu().numpy()
    want = o[dst]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=LAYER_ATOL * max(float(np.abs(want).max()), 1.0))
    assert rel_l1(got, want) < 2e-6, rel_l1(got, want)


def test_cfg2_cost_volume_and_maps_match_oracle(cfg2):
    o = cfg2["o"]
    D, h, w = o["cost"].shape
    ws = _lib.alloc_workspace(1, 32, D, h, w, DEV)
    cost = _lib.costreg_forward(_lib.to_c8(cu(o["var"])), cfg2["blob"], ws).cpu().numpy()
    np.testing.assert_allclose(cost, o["cost"], rtol=0, atol=3e-4 * max(float(np.abs(o["cost"]).max()), 1.0))
    # whole path, one call; measured 2.9e-7 -- the bound leaves one order of magnitude, not three
    N = cfg2["feats"].shape[0]
    ws = _lib.alloc_workspace(N, 32, D, h, w, DEV)
    depth = torch.empty((h, w), dtype=torch.float32, device=DEV)
    conf = torch.empty_like(depth)
    _lib.depth_infer(cu(cfg2["feats"]), cu(cfg2["proj"]), cu(cfg2["dv"]), cfg2["blob"], ws, depth, conf)
    r = rel_l1(depth.cpu().numpy(), o["depth"])
    assert r < 5e-6, r                                    # north_star bound: 1e-3
    assert (np.abs(conf.cpu().numpy() - o["conf"]) > 5e-3).mean() < 0.01


def test_cfg2_maps_match_the_reference_fixture(cfg2):
    """bench.py's workload against maps produced by the imported REFERENCE itself (tests/golden/fx_cfg2_maps.npz,
    made by tests/golden/gen_golden_cfg2.py with the reference's homo_warping / CostRegNet / depth_regression):
    the number bench.py times is pinned to the reference directly, not only through the oracle."""
    fx = load_fixture("cfg2_maps")
    N, _, h, w = cfg2["feats"].shape
    D = cfg2["dv"].shape[0]
    assert abs(float(np.abs(cfg2["feats"].astype(np.float64)).sum()) / float(fx["feats_checksum"]) - 1) < 1e-6
    ws = _lib.alloc_workspace(N, 32, D, h, w, DEV)
    depth = torch.empty((h, w), dtype=torch.float32, device=DEV)
    conf = torch.empty_like(depth)
    _lib.depth_infer(cu(cfg2["feats"]), cu(cfg2["proj"]), cu(cfg2["dv"]), cfg2["blob"], ws, depth, conf)
    r = rel_l1(depth.cpu().numpy(), fx["depth"])
    print(f"[cfg2] depth rel-L1 vs the reference fixture = {r:.3e}")
    assert r < 5e-6, r                                    # north_star bound: 1e-3
    assert_conf_close(conf.cpu().numpy(), fx["photometric_confidence"], fx["expected_index"], atol=5e-4)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_cfg2_two_maps_in_flight_are_bit_identical_to_one(cfg2, storage):
    """bench.py's `value` mode: two maps in flight on two HIP streams, each with its own workspace.  Every map must be
    the single-stream map bit for bit -- kernels of one map run BESIDE kernels of the other on the same CUs.  (Round 4:
    a packed-fp32 instruction of the warp kernel whose destination aliased its op_sel-swizzled weight operand gave a
    wrong 16-lane pass now and then, and only beside a bf16-MFMA kernel of the other stream; single-stream parity
    tests cannot see that.)  Two host threads, two different problems, the bench size."""
    import threading
    code = _lib.dtype_code(storage)
    c = synthetic.CONFIGS["cfg2"]
    N, h, w, D = c["nviews"], c["H"] // 4, c["W"] // 4, c["D"]
    probs = [(cu(cfg2["feats"]), cu(cfg2["proj"]), cu(cfg2["dv"])),
             (cu(synthetic.random_features(N, 32, h, w, seed=4)), cu(synthetic.cameras(N, h, w, yaw_deg=0.5)), cu(cfg2["dv"]))]

    def run(i, ws):
        depth = torch.empty((h, w), device=DEV)
        conf = torch.empty_like(depth)
        _lib.depth_infer(*probs[i], cfg2["blob"], ws, depth, conf, dtype=code)
        return depth, conf

    want = [run(i, _lib.alloc_workspace(N, 32, D, h, w, DEV, code)) for i in range(2)]
    torch.cuda.synchronize()
    got, errors = [[], []], []

    def worker(i):
        try:
            st = torch.cuda.Stream(DEV)
            with torch.cuda.stream(st):
                ws = _lib.alloc_workspace(N, 32, D, h, w, DEV, code)
                for _ in range(25):
                    got[i].append(run(i, ws))
            st.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    bad = [(i, j) for i in range(2) for j, (d, cf) in enumerate(got[i])
           if not (torch.equal(d, want[i][0]) and torch.equal(cf, want[i][1]))]
    assert not bad, f"{len(bad)} of 50 maps differ from the single-stream result: {bad[:8]}"


def test_cfg2_fused_conv11_prob_matches_oracle(cfg2):
    """mvs_conv11_prob (conv11 + conv0 skip + prob in ONE kernel, the default tail of the fp32 path) on its own
    at the bench size: fed with the oracle's d9 and c0, compared with the oracle's logits
    (models/mvsnet.py:71-72).  The kernel leans on one wave's LDS operations executing in order."""
    o = cfg2["o"]
    cost = _lib.conv11_prob(_lib.to_c8(cu(o["d9"])), _lib.to_c8(cu(o["c0"])), cfg2["blob"]).cpu().numpy()
    want = o["cost"]
    assert cost.shape == want.shape
    np.testing.assert_allclose(cost, want, rtol=0, atol=LAYER_ATOL * max(float(np.abs(want).max()), 1.0))
    assert rel_l1(cost, want) < 2e-6, rel_l1(cost, want)
    again = _lib.conv11_prob(_lib.to_c8(cu(o["d9"])), _lib.to_c8(cu(o["c0"])), cfg2["blob"]).cpu().numpy()
    assert np.array_equal(cost, again)                    # no race: bit-identical from run to run


@pytest.mark.parametrize("layer,shape", [(0, (32, 48, 64, 96)), (2, (16, 32, 48, 64)), (4, (32, 16, 24, 32))])
def test_winograd_layers_on_a_heavy_tailed_nonnegative_volume(cfg2, layer, shape):
    """conv0 (Winograd F(4,3) along z: constants 4, 5, 8, 1/6, 1/24) and conv2 / conv4 (F(2,3)) on an input shaped
    like a variance volume of trained features -- non-negative, heavy-tailed (exp(3 N(0,1)): dynamic range
    > 1e4) -- where the transforms' cancellation error is relative to the LARGEST neighbour, not to the voxel.
    Bound: |err| <= 2.5e-7 (4 ulp) x the maximum of |input| over the 9x3x3 neighbourhood that can reach the output
    through the transform x the layer's weight mass (sum |w| per output channel); measured: conv0 4e-8, conv2 1.6e-8,
    conv4 1.2e-8 of that scale (rel-L1 9e-7 / 4e-7 / 5e-7) -- F(4,3) along z costs no accuracy on such data."""
    sd = cfg2["sd"]
    C, D, h, w = shape
    g = np.random.default_rng(77 + layer)
    x = np.exp(3.0 * g.standard_normal((C, D, h, w))).astype(np.float32)
    assert x.max() / np.median(x) > 1e4
    name = f"conv{layer}"
    want = orc.conv3d(x, sd[f"{name}.conv.weight"], bn=orc._bn(sd, f"{name}.bn"))
    got = _lib.from_c8(_lib.conv_layer(layer, _lib.to_c8(cu(x)), None, cfg2["blob"])).cpu().numpy()
    # local scale: max |x| over the channel axis and a +-4 (z: an F(4,3) tile reads 6 planes for 4 outputs) x +-1 x +-1 window
    m = torch.from_numpy(x.max(axis=0))[None, None]
    m = torch.nn.functional.max_pool3d(m, kernel_size=(9, 3, 3), stride=1, padding=(4, 1, 1))[0, 0].numpy()
    gamma, beta, mean, var_ = orc._bn(sd, f"{name}.bn")
    wmass = (np.abs(sd[f"{name}.conv.weight"]).reshape(want.shape[0], -1).sum(1) * np.abs(gamma) / np.sqrt(var_ + 1e-5))
    bound = 2.5e-7 * wmass[:, None, None, None] * m[None] + 1e-6
    err = np.abs(got - want)
    worst = float((err / bound).max())
    print(f"[{name} heavy-tailed] max err / bound = {worst:.3f}, rel-L1 = {rel_l1(got, want):.2e}")
    assert worst <= 1.0, worst
    assert rel_l1(got, want) < 5e-6


@pytest.mark.parametrize("split", ["0", "2"])
def test_cfg2_conv0_kernel_variants_match_oracle(split):
    """conv0's other kernels at FULL size with the per-layer bounds of test_cfg2_every_layer_matches_oracle[0]: the
    fp32-MFMA Winograd kernel (MVS_CONV0_SPLIT=0; the default until round 4) and the first form of the split-operand
    kernel (=2).  Selection is read once per process -> child process (tests/conv0_check.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "conv0_check.py")], env=dict(os.environ, MVS_CONV0_SPLIT=split),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


def test_cfg2_layer_test_would_catch_a_dropped_tap(cfg2):
    """The sensitivity the per-layer comparison buys: zero ONE of conv4's 27x32x32 taps in the
    oracle and the per-layer metric moves far beyond its bound, while end-to-end depth moves 1e-5."""
    o, sd = cfg2["o"], dict(cfg2["sd"])
    w = sd["conv4.conv.weight"].copy()
    w[:, :, 1, 1, 1] = 0.0
    broken = orc.conv3d(o["c3"], w, bn=orc._bn(sd, "conv4.bn"))
    assert rel_l1(broken, o["c4"]) > 1e-3
    assert np.abs(broken - o["c4"]).max() > 50 * LAYER_ATOL * max(float(np.abs(o["c4"]).max()), 1.0)


# ------------------------------------------------------------------ cfg3 / cfg5 (BASELINE configs 2 / 4)
_FP32_ORACLE = {}   # cfg -> (var32, d32): the cfg3 oracle run (~20 s, 3.9 GB) is shared by its bf16 and fp32 cases


@pytest.mark.parametrize("cfg_name,storage", [("cfg5", "f16"), ("cfg3", "bf16"), ("cfg3", "f32")])
def test_16bit_configs_per_stage_and_against_the_fp32_oracle(cfg_name, storage):
    """configs[4] (N=4, fp16) and configs[2] (N=5, 1600x1184, D=256, bf16) at full size -- and configs[2]'s shape with
    fp32 volumes, the largest fp32 problem of the suite (3.88 GB variance volume: the 31-bit-offset guards of the
    conv kernels, the raw-buffer stores of the warp kernel beyond 2 GiB, and the fp32 kernels that only large
    shapes select -- depth-slab-fastest warp order, z-marching conv1, the chunk split of the fused tail):
      * the variance volume against the oracle with the same rounding points -- at cfg3 this is the
        depth-slab-fastest block order of the warp kernels, which small shapes never select;
      * depth against the rounding-matched oracle (tight) AND against the plain fp32 oracle, where
        north_star's 1e-3 relative L1 is asserted for the 16-bit storage modes too."""
    c = synthetic.CONFIGS[cfg_name]
    N, h, w, D = c["nviews"], c["H"] // 4, c["W"] // 4, c["D"]
    feats = synthetic.random_features(N, 32, h, w, seed=21)
    proj = synthetic.cameras(N, h, w, yaw_deg=0.5)
    dv = synthetic.depth_values(D, interval_scale=c["interval_scale"])
    sd = synthetic.random_costreg_state(seed=2)
    code = _lib.dtype_code(storage)

    if cfg_name not in _FP32_ORACLE:
        _FP32_ORACLE.clear()
        v = orc.variance_volume(feats, proj, dv)                       # the fp32 reference volume
        _FP32_ORACLE[cfg_name] = (v, orc.softargmin_conf(orc.costreg_forward(v, sd), dv)[0])
    var32, d32 = _FP32_ORACLE[cfg_name]
    var_m = var32 if storage == "f32" else orc.round_storage(
        orc.variance_volume(feats, proj, dv), storage)
    got = _lib.from_c8(hip_variance_c8(feats, proj, dv, code).float()).cpu().numpy()
    eps = {"f32": 0.0, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}[storage]
    np.testing.assert_allclose(got, var_m, rtol=eps, atol=5e-4)
    del got

    ws = _lib.alloc_workspace(N, 32, D, h, w, DEV, code)
    depth = torch.empty((h, w), dtype=torch.float32, device=DEV)
    conf = torch.empty_like(depth)
    _lib.depth_infer(cu(feats), cu(proj), cu(dv), _lib.pack_weights(sd).to(DEV), ws, depth, conf, dtype=code)
    depth = depth.cpu().numpy()
    assert np.isfinite(depth).all()
    r32 = rel_l1(depth, d32)
    print(f"[{cfg_name} {storage}] depth rel-L1 vs the fp32 oracle = {r32:.3e}")
    if storage == "f32":
        assert r32 < 5e-6, r32
        _FP32_ORACLE.clear()
        return
    d_m, _, _ = orc.softargmin_conf(orc.costreg_forward(var_m, sd, storage, arith16=True), dv)
    rm = rel_l1(depth, d_m)
    print(f"[{cfg_name} {storage}] depth rel-L1 vs the rounding-matched oracle = {rm:.3e}")
    assert rm < (2e-4 if storage == "f16" else 1e-3), rm
    assert r32 < 1e-3, r32                                  # north_star, against the fp32 path
