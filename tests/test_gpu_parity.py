"""GPU parity tests: the HIP path (through the C ABI, ctypes) against
  (1) golden vectors captured from the imported reference (tests/golden/fx_*.npz), and
  (2) the CPU oracle (oracle/) on seeded inputs,
stage by stage and end to end.  Tolerances are fp32 summation-order noise; the north_star
bound (1e-3 relative L1 on depth) is asserted end to end with orders of magnitude to spare.
"""
import numpy as np
import pytest
import torch

from conftest import assert_conf_close, load_fixture, load_weights, rel_l1
from oracle import oracle as orc
from scene_3dreconstruction_mvsnet_amd import MVSNet, _lib, synthetic
from scene_3dreconstruction_mvsnet_amd import module as hip_module

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def costreg_sd(fx=None):
    sd = orc.costreg_state(load_weights())
    if fx is not None and "prob_gain_extra" in fx:
        sd = dict(sd)
        sd["prob.weight"] = sd["prob.weight"] * fx["prob_gain_extra"]
    return sd


def blob_for(sd):
    return _lib.pack_weights(sd).to(DEV)


def hip_variance(feats, proj, dv):
    """-> numpy [C,D,h,w] (reference layout) from the private C8-planar volume."""
    N, C, h, w = feats.shape
    D = dv.shape[0]
    ws = _lib.alloc_workspace(N, C, D, h, w, DEV)
    rt = _lib.relative_proj(cu(proj))
    var = _lib.warp_variance(cu(feats), rt, cu(dv), ws)
    torch.cuda.synchronize()
    return _lib.from_c8(var).cpu().numpy()


def hip_costreg(var_ncdhw, sd):
    C, D, h, w = var_ncdhw.shape
    var = _lib.to_c8(cu(var_ncdhw))
    ws = _lib.alloc_workspace(1, C, D, h, w, DEV)
    cost = _lib.costreg_forward(var, blob_for(sd), ws)
    torch.cuda.synchronize()
    return cost.cpu().numpy()


def hip_depth_infer(feats, proj, dv, sd, dtype=_lib.MVS_F32):
    N, C, h, w = feats.shape
    D = dv.shape[0]
    ws = _lib.alloc_workspace(N, C, D, h, w, DEV, dtype)
    depth = torch.empty((h, w), dtype=torch.float32, device=DEV)
    conf = torch.empty_like(depth)
    _lib.depth_infer(cu(feats), cu(proj), cu(dv), blob_for(sd), ws, depth, conf, dtype=dtype)
    torch.cuda.synchronize()
    return depth.cpu().numpy(), conf.cpu().numpy()


# ------------------------------------------------------------------------------ per-stage parity
@pytest.mark.parametrize("name", ["tiny", "small", "n5yaw", "oob"])
def test_relative_proj(name):
    fx = load_fixture(name)
    proj = fx["proj_matrices"][0]
    rt = _lib.relative_proj(cu(proj)).cpu().numpy()
    for v in range(1, proj.shape[0]):
        want = orc.relative_proj(proj[v], proj[0])
        np.testing.assert_allclose(rt[v - 1], want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", ["tiny", "oob"])
def test_homo_warping_matches_reference(name):
    fx = load_fixture(name)
    feats, proj, dv = fx["features"], fx["proj_matrices"], fx["depth_values"]
    for v in range(1, feats.shape[1]):
        got = hip_module.homo_warping(cu(feats[:, v]), cu(proj[:, v]), cu(proj[:, 0]), cu(dv))
        assert got.shape == fx["warped"][:, v - 1].shape
        np.testing.assert_allclose(got.cpu().numpy(), fx["warped"][:, v - 1], rtol=0, atol=3e-4)


@pytest.mark.parametrize("name", ["tiny", "small", "n5yaw", "oob", "sharp"])
def test_warp_variance_matches_reference(name):
    fx = load_fixture(name)
    got = hip_variance(fx["features"][0], fx["proj_matrices"][0], fx["depth_values"][0])
    np.testing.assert_allclose(got, fx["variance"][0], rtol=0, atol=5e-4)


@pytest.mark.parametrize("name", ["tiny", "small", "n5yaw", "sharp"])
def test_costreg_matches_reference(name):
    fx = load_fixture(name)
    got = hip_costreg(fx["variance"][0], costreg_sd(fx))
    scale = max(np.abs(fx["cost_reg"][0]).max(), 1.0)
    np.testing.assert_allclose(got, fx["cost_reg"][0], rtol=0, atol=3e-4 * scale)


@pytest.mark.parametrize("name", ["tiny", "small", "sharp", "cfg1"])
def test_softargmin_conf_matches_reference(name):
    fx = load_fixture(name)
    depth, conf = _lib.softargmin_conf(cu(fx["cost_reg"][0]), cu(fx["depth_values"][0]))
    depth, conf = depth.cpu().numpy(), conf.cpu().numpy()
    np.testing.assert_allclose(depth, fx["depth"][0], rtol=0, atol=3e-3)  # ~450 mm scale
    assert_conf_close(conf, fx["photometric_confidence"][0], fx["expected_index"][0],
                      prob=fx["prob_volume"][0] if "prob_volume" in fx else None, atol=2e-5)


def test_depth_regression_matches_reference():
    fx = load_fixture("small")
    got = hip_module.depth_regression(cu(fx["prob_volume"]), cu(fx["depth_values"]))
    np.testing.assert_allclose(got.cpu().numpy(), fx["depth"], rtol=0, atol=3e-3)


@pytest.mark.parametrize("D,h,w", [(8, 8, 8), (8, 16, 40), (16, 24, 72), (10, 8, 96)])
def test_conv0_mfma_matches_oracle(D, h, w):
    """The default fp32-MFMA conv0 kernel (Winograd F(4,3) along z, ragged x tiles) against the oracle."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((32, D, h, w)).astype(np.float32)
    sd = synthetic.random_costreg_state(seed=9)
    want = orc.conv3d(x, sd["conv0.conv.weight"], bn=orc._bn(sd, "conv0.bn"))
    got = _lib.from_c8(_lib.conv_layer(0, _lib.to_c8(cu(x)), None, blob_for(sd))).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4 * max(np.abs(want).max(), 1.0))


@pytest.mark.parametrize("D,h,w", [(8, 8, 8), (8, 16, 24), (16, 32, 40), (24, 40, 72), (40, 56, 64), (16, 128, 8)])
def test_fused_conv11_prob_matches_the_two_launches(D, h, w):
    """mvs_conv11_prob (the tail of mvs_costreg_forward for fp32 storage) against mvs_conv_layer 9 and 10 as
    separate kernels.  Same taps, different summation order in the stencil: equal to fp32 rounding, on
    shapes with ragged tiles in y / x, one to three z chunks and tiles narrower than a block."""
    sd = synthetic.random_costreg_state(seed=21)
    blob = blob_for(sd)
    g = torch.Generator(device="cpu").manual_seed(D * 1000 + h)
    var = (torch.rand((4, D, h, w, 8), generator=g) * 0.5).to(DEV)
    ws = _lib.alloc_workspace(2, 32, D, h, w, DEV)
    a = [None] * 10
    a[0] = _lib.conv_layer(0, var, None, blob)
    for l in range(1, 7):
        a[l] = _lib.conv_layer(l, a[l - 1], None, blob)
    a[7] = _lib.conv_layer(7, a[6], a[4], blob)
    a[8] = _lib.conv_layer(8, a[7], a[2], blob)
    a[9] = _lib.conv_layer(9, a[8], a[0], blob)
    want = _lib.conv_layer(10, a[9], None, blob)
    pair = _lib.conv11_prob(a[8], a[0], blob)
    whole = _lib.costreg_forward(var, blob, ws)
    torch.cuda.synchronize()
    want, pair, whole = want.cpu().numpy(), pair.cpu().numpy(), whole.cpu().numpy()
    assert np.isfinite(pair).all()
    np.testing.assert_allclose(pair, want, rtol=0, atol=2e-6 * max(np.abs(want).max(), 1.0))
    np.testing.assert_array_equal(whole, pair)   # mvs_costreg_forward runs the same kernels on the same inputs


@pytest.mark.parametrize("storage", ["f16", "bf16"])
@pytest.mark.parametrize("D,h,w", [(8, 8, 8), (16, 24, 40), (24, 16, 72)])
def test_fused_conv11_prob_16bit_matches_oracle(storage, D, h, w):
    """mvs_conv11_prob with 16-bit storage (transposed convolution on the 16-bit MFMA, prob stencil in fp32 on the
    never-stored sum): against the oracle with the same rounding points -- x, skip and the BN-folded conv11 weights
    rounded to the storage dtype, fp32 accumulation, conv11 + skip NOT rounded, fp32 prob (models/mvsnet.py:71-72)
    -- on shapes with ragged tiles and several z chunks; and against the two 16-bit launches it replaces, which
    differ from it only by the rounding of the tensor in between."""
    code = _lib.dtype_code(storage)
    tdt = _lib.TORCH_DTYPES[code]
    q = lambda t: orc.round_storage(t, storage)  # noqa: E731
    sd = synthetic.random_costreg_state(seed=23)
    blob = blob_for(sd)
    rng = np.random.default_rng(D * 100 + w)
    x = q(np.abs(rng.standard_normal((16, D // 2, h // 2, w // 2))).astype(np.float32))
    skip = q(np.abs(rng.standard_normal((8, D, h, w))).astype(np.float32))
    wf, sh = orc._fold(sd, "conv11.0.weight", "conv11.1", transposed=True)
    wt = np.ascontiguousarray(q(wf).transpose(1, 0, 2, 3, 4))
    d11 = skip + np.maximum(orc.deconv3d(x, wt, bn=None, relu=False) + sh[:, None, None, None], 0.0)
    want = orc.conv3d(d11, sd["prob.weight"], bias=sd["prob.bias"], bn=None, relu=False)[0]
    xt, st = _lib.to_c8(cu(x)).to(tdt), _lib.to_c8(cu(skip)).to(tdt)
    got = _lib.conv11_prob(xt, st, blob, dtype=code).cpu().numpy()
    assert np.isfinite(got).all()
    scale = max(float(np.abs(want).max()), 1.0)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-5 * scale)    # fp32 summation order only
    two = _lib.conv_layer(10, _lib.conv_layer(9, xt, st, blob, dtype=code), None, blob, dtype=code).cpu().numpy()
    eps = {"f16": 2.0 ** -10, "bf16": 2.0 ** -7}[storage]
    np.testing.assert_allclose(got, two, rtol=0, atol=4 * eps * scale)  # the rounding of d11 the fused form skips
    assert np.array_equal(got, _lib.conv11_prob(xt, st, blob, dtype=code).cpu().numpy())   # run-to-run identical


@pytest.mark.parametrize("layer", list(range(11)))
def test_every_layer_matches_oracle(layer):
    """mvs_conv_layer for each CostRegNet layer on random C8-planar input vs the oracle."""
    ci, co = _lib._LAYER_CH[layer]
    rng = np.random.default_rng(layer)
    D, h, w = (8, 8, 16)
    x = rng.standard_normal((ci, D, h, w)).astype(np.float32)
    sd = synthetic.random_costreg_state(seed=3)
    blob = blob_for(sd)
    key = _lib.CONV_WEIGHT_KEYS[layer]
    if layer == 10:
        want = orc.conv3d(x, sd[key], bias=sd["prob.bias"], bn=None, relu=False)[0]
        got = _lib.conv_layer(10, _lib.to_c8(cu(x)), None, blob).cpu().numpy()
    elif layer >= 7:
        skip = rng.standard_normal((co, 2 * D, 2 * h, 2 * w)).astype(np.float32)
        want = skip + orc.deconv3d(x, sd[key], bn=orc._bn(sd, _lib.BN_PREFIXES[layer]))
        got = _lib.from_c8(_lib.conv_layer(layer, _lib.to_c8(cu(x)), _lib.to_c8(cu(skip)), blob)).cpu().numpy()
    else:
        stride = 2 if layer in (1, 3, 5) else 1
        want = orc.conv3d(x, sd[key], bn=orc._bn(sd, _lib.BN_PREFIXES[layer]), stride=stride)
        got = _lib.from_c8(_lib.conv_layer(layer, _lib.to_c8(cu(x)), None, blob)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4 * max(np.abs(want).max(), 1.0))


# ------------------------------------------------------------------------------ end to end
@pytest.mark.parametrize("name", ["tiny", "small", "n5yaw", "oob", "b2", "cfg1", "sharp"])
def test_depth_infer_matches_reference(name):
    fx = load_fixture(name)
    sd = costreg_sd(fx)
    for b in range(fx["features"].shape[0]):
        depth, conf = hip_depth_infer(fx["features"][b], fx["proj_matrices"][b],
                                      fx["depth_values"][b], sd)
        assert rel_l1(depth, fx["depth"][b]) < 1e-5  # north_star: 1e-3
        assert_conf_close(conf, fx["photometric_confidence"][b], fx["expected_index"][b],
                          prob=fx["prob_volume"][b] if "prob_volume" in fx else None, atol=1e-3)


@pytest.mark.parametrize("name", ["tiny", "small", "b2"])
def test_mvsnet_module_forward_from_images(name):
    """Drop-in module: images in, dict out (FeatureNet on PyTorch-ROCm + HIP path)."""
    fx = load_fixture(name)
    w = load_weights()
    model = torch.nn.DataParallel(MVSNet(refine=False), device_ids=[0]).to(DEV)
    model.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in w.items()})
    model.eval()
    out = model(cu(fx["imgs"]), cu(fx["proj_matrices"]), cu(fx["depth_values"]))
    assert set(out.keys()) == {"depth", "photometric_confidence"}
    assert out["depth"].dtype == torch.float32 and out["depth"].device.type == "cuda"
    assert tuple(out["depth"].shape) == fx["depth"].shape
    assert rel_l1(out["depth"].cpu().numpy(), fx["depth"]) < 1e-4
    conf = out["photometric_confidence"].cpu().numpy()
    frac_bad = (np.abs(conf - fx["photometric_confidence"]) > 5e-3).mean()
    assert frac_bad < 0.02


def test_weight_update_invalidates_blob_cache():
    fx = load_fixture("tiny")
    w = load_weights()
    model = MVSNet(refine=False).to(DEV).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    args = (cu(fx["imgs"]), cu(fx["proj_matrices"]), cu(fx["depth_values"]))
    d0 = model(*args)["depth"].clone()
    with torch.no_grad():
        model.cost_regularization.prob.weight.mul_(3.0)
    d1 = model(*args)["depth"]
    assert (d0 - d1).abs().max() > 1e-3
    with torch.no_grad():
        model.cost_regularization.prob.weight.div_(3.0)
    d2 = model(*args)["depth"]
    assert rel_l1(d2.cpu().numpy(), d0.cpu().numpy()) < 1e-6


# ------------------------------------------------------------------------------ vs oracle, seeded
@pytest.mark.parametrize("N,h,w,D,kw", [
    (2, 8, 8, 8, {}),
    (4, 24, 40, 32, dict(yaw_deg=2.0)),
    (5, 32, 48, 24, dict(baseline=(-45.0, 12.0, 3.0))),
    (7, 16, 16, 16, dict(baseline=(-12.0, 3.0, 1.0))),
    (1, 16, 16, 8, {}),                      # single view: variance == 0 everywhere
])
def test_depth_infer_matches_oracle_random(N, h, w, D, kw):
    feats = synthetic.random_features(N, 32, h, w, seed=11)
    proj = synthetic.cameras(N, h, w, **kw)
    dv = synthetic.depth_values(D)
    sd = synthetic.random_costreg_state(seed=4)
    var = orc.variance_volume(feats, proj, dv)
    np.testing.assert_allclose(hip_variance(feats, proj, dv), var, rtol=0, atol=5e-4)
    cost = orc.costreg_forward(var, sd)
    np.testing.assert_allclose(hip_costreg(var, sd), cost, rtol=0,
                               atol=3e-4 * max(np.abs(cost).max(), 1.0))
    depth_o, conf_o, idx_o, prob_o = orc.softargmin_conf(cost, dv, want_prob=True)
    depth, conf = hip_depth_infer(feats, proj, dv, sd)
    assert rel_l1(depth, depth_o) < 1e-5
    assert_conf_close(conf, conf_o, idx_o, prob=prob_o, atol=1e-3)


@pytest.mark.parametrize("D,h,w,gain", [(8, 5, 7, 1.0), (48, 12, 20, 3.0), (192, 16, 24, 1.0), (256, 8, 40, 10.0),
                                         (272, 8, 24, 3.0), (17, 3, 5, 30.0)])
def test_softargmin_conf_random_logits(D, h, w, gain):
    """mvs_softargmin_conf on random logits against the oracle (models/mvsnet.py:192-218): the register-resident
    form (D <= 128: 8 logits per thread, D <= 256: 16) and the looping form (D = 272), ragged pixel blocks (h*w not
    a multiple of 16), D not a multiple of the 16 slices, sharp and flat distributions; the confidence window
    straddling two slices is summed through LDS."""
    rng = np.random.default_rng(D * 31 + w)
    cost = (gain * rng.standard_normal((D, h, w))).astype(np.float32)
    dv = synthetic.depth_values(D)
    depth_o, conf_o, idx_o, prob_o = orc.softargmin_conf(cost, dv, want_prob=True)
    depth, conf = _lib.softargmin_conf(cu(cost), cu(dv))
    depth, conf = depth.cpu().numpy(), conf.cpu().numpy()
    assert rel_l1(depth, depth_o) < 2e-6
    assert_conf_close(conf, conf_o, idx_o, prob=prob_o, atol=2e-5)


def test_nonfinite_coordinates_give_nan_like_torch():
    C, h, w = 32, 8, 8
    fea = np.ones((1, C, h, w), np.float32)
    ref = np.eye(4, dtype=np.float32)[None]
    src = np.eye(4, dtype=np.float32)[None]
    src[0, 2, 3] = -1.0
    dv = np.array([[1.0, 2.0]], np.float32)
    out = hip_module.homo_warping(cu(fea), cu(src), cu(ref), cu(dv)).cpu().numpy()
    want = orc.homo_warp(fea[0], src[0], ref[0], dv[0])
    assert np.isnan(out[0, :, 0]).all() and np.isfinite(out[0, :, 1]).all()
    np.testing.assert_allclose(out[0][:, 1], want[:, 1], atol=1e-5)


@pytest.mark.parametrize("env", [
    {"MVS_WARP_TC": "0"},        # plain gather warp+variance kernel (tap cache off; also the N > 5 path)
    {"MVS_CONV0_WINO": "0", "MVS_CONV_WINO": "0"},   # direct MFMA kernels (no Winograd transform anywhere)
    {"MVS_CONV0_SPLIT": "0"},    # conv0 on the fp32 MFMA (Winograd F(4,3)) instead of split bf16 operands
    {"MVS_SPLIT_LAYERS": "0"},   # conv2 .. conv4 on the fp32 MFMA (Winograd F(2,3) / generic kernels)
    {"MVS_CONV0_SPLIT": "2"},    # the split-operand conv0 in its first form (one tile per 4-wave block)
    {"MVS_FORCE_DIRECT": "1"},   # VALU direct convolutions for every layer
    {"MVS_FUSE_PROB": "0"},      # conv11 and prob as two launches
    {"MVS_TAIL_SPLIT": "0"},     # the fused tail's transposed convolution on the fp32 MFMA (conv11_prob_priv)
    {"MVS_SPLIT_DECONV": "3"},   # conv7 too with split operands (deconvgs; default: conv9 only)
    {"MVS_SPLIT_DECONV": "0"},   # conv7 / conv9 on the fp32 MFMA
])
def test_optin_kernel_variants(env):
    """The non-default kernels stay parity-green (selection is read once per process, so each
    variant runs tests/variant_check.py in a child process; sequential, one GPU user at a time)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    child_env = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(here, "variant_check.py")], env=child_env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_packed_tap_cache_is_bit_identical():
    """MVS_WARP_PACKED=1 (round-4 experiment): the 16-bit taps stay packed in the register cache and are
    widened inside the blend -- the variance volume must be the default kernel's, bit for bit, for fp16 and bf16."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, hashlib, torch; sys.path.insert(0, %r)\n"
        "from scene_3dreconstruction_mvsnet_amd import _lib, synthetic\n"
        "for st in ('f16', 'bf16'):\n"
        "    dt = _lib.dtype_code(st); N, h, w, D = 5, 40, 56, 48\n"
        "    f = torch.from_numpy(synthetic.random_features(N, 32, h, w, seed=3)).cuda()\n"
        "    p = torch.from_numpy(synthetic.cameras(N, h, w, yaw_deg=1.0)).cuda()\n"
        "    dv = torch.from_numpy(synthetic.depth_values(D)).cuda()\n"
        "    ws = _lib.alloc_workspace(N, 32, D, h, w, 'cuda:0', dt)\n"
        "    v = _lib.warp_variance(f, _lib.relative_proj(p), dv, ws, dtype=dt)\n"
        "    print(st, hashlib.sha1(v.cpu().view(torch.int16).numpy().tobytes()).hexdigest())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    outs = []
    for pk in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MVS_WARP_PACKED=pk), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] and outs[0].count("\n") == 2, outs


@pytest.mark.parametrize("storage", ["f16", "bf16"])
def test_16bit_storage_16bit_feature_copy_variant(storage):
    """MVS_FEAT16=1: the 16-bit modes gather from a feature copy narrowed to the storage type (the default of rounds
    2-3; since round 4 the gather reads the fp32 copy); against the oracle with the same rounding point."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import load_fixture, load_weights, rel_l1\n"
        "from oracle import oracle as orc\n"
        "from scene_3dreconstruction_mvsnet_amd import _lib\n"
        "fx = load_fixture('n5yaw'); sd = orc.costreg_state(load_weights()); st = %r\n"
        "f, p, d = fx['features'][0], fx['proj_matrices'][0], fx['depth_values'][0]\n"
        "dev = 'cuda:0'; cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)\n"
        "N, C, h, w = f.shape; dt = _lib.dtype_code(st)\n"
        "ws = _lib.alloc_workspace(N, C, d.shape[0], h, w, dev, dt)\n"
        "depth = torch.empty((h, w), device=dev); conf = torch.empty_like(depth)\n"
        "_lib.depth_infer(cu(f), cu(p), cu(d), _lib.pack_weights(sd).to(dev), ws, depth, conf, dtype=dt)\n"
        "want, _ = orc.depth_infer(f, p, d, sd, storage=st, feat16=True)\n"
        "r = rel_l1(depth.cpu().numpy(), want); print(r); sys.exit(0 if r < (2e-4 if st == 'f16' else 1e-3) else 1)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), storage)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MVS_FEAT16="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("storage", ["f16", "bf16"])
def test_16bit_storage_fp32_arithmetic_variant(storage):
    """MVS_MFMA16=0: fp32 MFMA on the narrowed operands, against the oracle that rounds storage only."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import load_fixture, load_weights, rel_l1\n"
        "from oracle import oracle as orc\n"
        "from scene_3dreconstruction_mvsnet_amd import _lib\n"
        "fx = load_fixture('small'); sd = orc.costreg_state(load_weights()); st = %r\n"
        "f, p, d = fx['features'][0], fx['proj_matrices'][0], fx['depth_values'][0]\n"
        "dev = 'cuda:0'; cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)\n"
        "N, C, h, w = f.shape; dt = _lib.dtype_code(st)\n"
        "ws = _lib.alloc_workspace(N, C, d.shape[0], h, w, dev, dt)\n"
        "depth = torch.empty((h, w), device=dev); conf = torch.empty_like(depth)\n"
        "_lib.depth_infer(cu(f), cu(p), cu(d), _lib.pack_weights(sd).to(dev), ws, depth, conf, dtype=dt)\n"
        "want, _ = orc.depth_infer(f, p, d, sd, storage=st, arith16=False)\n"
        "r = rel_l1(depth.cpu().numpy(), want); print(r); sys.exit(0 if r < (2e-4 if st == 'f16' else 1e-3) else 1)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), storage)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MVS_MFMA16="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("env,shape", [
    # persistent conv1 kernel (conv3d_mfma.hip: taken when tiles >= 4 x resident blocks, i.e. only at
    # full size by default): MVS_PERSIST_CUS=1 sizes the grid for "one CU" = 4 blocks; 16 and 72 tiles
    ({"MVS_PERSIST_CUS": "1"}, ("16", "16", "32", "f32", "f16")),
    ({"MVS_PERSIST_CUS": "1"}, ("24", "24", "40", "f32")),
    # depth-slab-fastest block order of the warp kernels (default only when the features exceed L2)
    ({"MVS_WARP_DEPTH_FASTEST": "1"}, ("24", "24", "40", "f32", "bf16")),
    ({"MVS_WARP_DEPTH_FASTEST": "1", "MVS_WARP_TC": "0"}, ("16", "16", "32", "f32")),
    ({"MVS_WARP_DEPTH_FASTEST": "1", "MVS_WARP_TC16": "0"}, ("16", "16", "32", "f16", "bf16")),   # plain 16-bit kernel
    ({"MVS_WARP_TC16": "0"}, ("16", "16", "32", "bf16")),
    # z-marching 16-bit conv0 (conv3d_mfma16.hip: default once its columns fill the chip): ragged x tile (w = 40),
    # several z chunks; and the tile kernel it replaces at full size
    ({"MVS_CONV0Z16": "1"}, ("24", "24", "40", "f16", "bf16")),
    ({"MVS_CONV0Z16": "1"}, ("16", "16", "32", "bf16")),
    ({"MVS_CONV0Z16": "0"}, ("16", "16", "32", "f16")),
    # z-marching 16-bit conv1 / conv2 / conv3 (conv3d_mfma16.hip convz16: default once their columns fill the chip),
    # and the tile kernels they replace at full size
    ({"MVS_CONVZ16": "1"}, ("24", "24", "40", "f16", "bf16")),
    ({"MVS_CONVZ16": "1"}, ("16", "16", "32", "bf16")),
    ({"MVS_CONVZ16": "0", "MVS_CONV0Z16": "0"}, ("16", "16", "32", "bf16")),
    # z-marching fp32 conv1 (conv3d_mfma.hip: default once its columns fill the chip): several z chunks with
    # surplus steps, ragged y / x tiles (h/2 = 12 rows for 8-row tiles, w/2 = 20 for 16-column tiles)
    ({"MVS_CONV1Z": "1"}, ("24", "24", "40", "f32")),
    ({"MVS_CONV1Z": "1"}, ("16", "16", "32", "f32")),
    ({"MVS_CONV1Z": "0"}, ("16", "16", "32", "f32")),
    # z-deep block tiles of the 16-bit tile kernels (conv4 - conv7, conv9: default where they still fill the chip), forced at
    # shapes with ragged tiles in every direction, and the round-2 tiles forced where the default takes the deep ones
    ({"MVS_DEEP_TILES": "1"}, ("24", "40", "56", "f16", "bf16")),
    ({"MVS_DEEP_TILES": "1"}, ("16", "24", "40", "bf16")),
    ({"MVS_DEEP_TILES": "0"}, ("16", "24", "40", "f16")),
])
def test_full_size_only_code_paths_at_small_shapes(env, shape):
    """Kernels / launch orders that the default selection reaches only at full size, forced at a
    small shape in a child process and compared per layer with the oracle (tests/layer_check.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "layer_check.py"), *shape],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


# ------------------------------------------------------------------------------ error behaviour
def test_bad_shapes_return_status_not_crash():
    feats = torch.zeros((3, 32, 12, 16), device=DEV)  # h=12 not a multiple of 8
    proj = torch.eye(4, device=DEV).repeat(3, 1, 1)
    dv = torch.linspace(425, 500, 8, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    depth = torch.empty((12, 16), device=DEV)
    with pytest.raises(_lib.MvsError) as e:
        _lib.depth_infer(feats, proj, dv, ws, ws, depth, depth.clone())
    assert e.value.code == 1
    feats = torch.zeros((3, 32, 16, 16), device=DEV)
    with pytest.raises(_lib.MvsError) as e:  # workspace too small
        _lib.depth_infer(feats, proj, dv, ws, ws[:1024], depth, depth.clone())
    assert e.value.code == 3


# ------------------------------------------------------------------------------ full size (cfg2)
@pytest.fixture(scope="module")
def cfg2_problem():
    c = synthetic.CONFIGS["cfg2"]
    N, h, w, D = c["nviews"], c["H"] // 4, c["W"] // 4, c["D"]
    feats = synthetic.random_features(N, 32, h, w, seed=0)
    proj = synthetic.cameras(N, h, w)
    dv = synthetic.depth_values(D, interval_scale=c["interval_scale"])
    sd = synthetic.random_costreg_state(seed=0)
    return feats, proj, dv, sd


def test_cfg2_full_size_matches_oracle(cfg2_problem):
    """BASELINE.json configs[1] (N=5, 640x512 -> 128x160, D=192) against the CPU oracle."""
    feats, proj, dv, sd = cfg2_problem
    depth, conf = hip_depth_infer(feats, proj, dv, sd)
    depth_o, conf_o = orc.depth_infer(feats, proj, dv, sd)
    assert np.isfinite(depth).all()
    assert rel_l1(depth, depth_o) < 5e-6  # measured 2.9e-7; north_star: 1e-3
    assert (np.abs(conf - conf_o) > 5e-3).mean() < 0.01


def test_cfg2_properties(cfg2_problem):
    """Size-independent properties at full size: determinism, range, source-view permutation
    invariance (variance is symmetric in the source views) and world-scale equivariance."""
    feats, proj, dv, sd = cfg2_problem
    d0, c0 = hip_depth_infer(feats, proj, dv, sd)
    d1, c1 = hip_depth_infer(feats, proj, dv, sd)
    assert np.array_equal(d0, d1) and np.array_equal(c0, c1)
    assert d0.min() >= dv[0] - 1e-3 and d0.max() <= dv[-1] + 1e-3
    assert c0.min() >= 0 and c0.max() <= 1 + 1e-5
    perm = [0, 3, 1, 4, 2]
    dp, _ = hip_depth_infer(feats[perm], proj[perm], dv, sd)
    assert rel_l1(dp, d0) < 1e-5
    # scale the world by s: translations and depth hypotheses scale, sampling is unchanged
    s = 2.0
    proj_s = proj.copy()
    proj_s[:, :3, 3] *= s
    ds, cs = hip_depth_infer(feats, proj_s, dv * s, sd)
    assert rel_l1(ds, d0 * s) < 1e-5
    assert (np.abs(cs - c0) > 1e-3).mean() < 0.01


# ------------------------------------------------------------------------------ 16-bit storage
@pytest.mark.parametrize("storage", ["f16", "bf16"])
@pytest.mark.parametrize("name", ["small", "n5yaw", "cfg1"])
def test_16bit_storage_matches_matched_oracle(name, storage):
    """fp16 / bf16 storage of the private volumes (fp32 arithmetic): against the oracle with the
    same rounding points, and -- for information and a loose bound -- against the fp32 reference."""
    fx = load_fixture(name)
    sd = costreg_sd(fx)
    code = _lib.dtype_code(storage)
    feats, proj, dv = fx["features"][0], fx["proj_matrices"][0], fx["depth_values"][0]
    depth, conf = hip_depth_infer(feats, proj, dv, sd, dtype=code)
    depth_m, conf_m = orc.depth_infer(feats, proj, dv, sd, storage=storage)
    assert np.isfinite(depth).all()
    # a few activations sit on a rounding boundary and flip with fp32 summation order, so the
    # match is close but not at fp32-noise level
    assert rel_l1(depth, depth_m) < (2e-4 if storage == "f16" else 1e-3)
    # vs the fp32 reference (north_star bound 1e-3 is for the fp32 path; recorded in DESIGN.md)
    assert rel_l1(depth, fx["depth"][0]) < (1e-3 if storage == "f16" else 1e-2)


@pytest.mark.parametrize("storage", ["f16", "bf16"])
def test_16bit_layers_match_matched_oracle(storage):
    """warp+variance and one layer of each kind with 16-bit storage vs rounded oracle outputs."""
    code = _lib.dtype_code(storage)
    tdt = _lib.TORCH_DTYPES[code]
    fx = load_fixture("small")
    feats, proj, dv = fx["features"][0], fx["proj_matrices"][0], fx["depth_values"][0]
    ws = _lib.alloc_workspace(feats.shape[0], 32, dv.shape[0], feats.shape[2], feats.shape[3], DEV, code)
    var = _lib.warp_variance(cu(feats), _lib.relative_proj(cu(proj)), cu(dv), ws, dtype=code)
    assert var.dtype == tdt
    # 16-bit modes: fp32 features, fp32 interpolation / variance, the volume narrowed in the store
    want = orc.round_storage(orc.variance_volume(feats, proj, dv), storage)
    eps = 2.0 ** (-10 if storage == "f16" else -7)
    got = _lib.from_c8(var.float()).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=eps, atol=5e-4)
    sd = synthetic.random_costreg_state(seed=3)
    blob = blob_for(sd)
    rng = np.random.default_rng(1)
    q = lambda t: orc.round_storage(t, storage)  # noqa: E731
    for layer in range(11):
        ci, co = _lib._LAYER_CH[layer]
        x = q(rng.standard_normal((ci, 8, 8, 16)).astype(np.float32))
        key = _lib.CONV_WEIGHT_KEYS[layer]
        xt = _lib.to_c8(cu(x)).to(tdt)
        if layer == 10:  # prob: fp32 weights, VALU
            want = orc.conv3d(x, sd[key], bias=sd["prob.bias"], bn=None, relu=False)[0]
            got = _lib.conv_layer(10, xt, None, blob, dtype=code).cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-4 * max(np.abs(want).max(), 1.0))
            continue
        # 16-bit MFMA mode: BN-folded weights rounded to the storage dtype, fp32 accumulation
        if layer >= 7:
            w, sh = orc._fold(sd, key, _lib.BN_PREFIXES[layer], transposed=True)
            wt = np.ascontiguousarray(q(w).transpose(1, 0, 2, 3, 4))
            skip = q(rng.standard_normal((co, 16, 16, 32)).astype(np.float32))
            want = skip + np.maximum(orc.deconv3d(x, wt, bn=None, relu=False) + sh[:, None, None, None], 0.0)
            y = _lib.conv_layer(layer, xt, _lib.to_c8(cu(skip)).to(tdt), blob, dtype=code)
        else:
            w, sh = orc._fold(sd, key, _lib.BN_PREFIXES[layer])
            stride = 2 if layer in (1, 3, 5) else 1
            want = orc.conv3d(x, q(w), bias=sh, bn=None, stride=stride, relu=True)
            y = _lib.conv_layer(layer, xt, None, blob, dtype=code)
        assert y.dtype == tdt
        got = _lib.from_c8(y.float()).cpu().numpy()
        # one storage ulp of slack on top of fp32 summation noise
        np.testing.assert_allclose(got, q(want), rtol=2 * eps, atol=3e-4 * max(np.abs(want).max(), 1.0))


# ------------------------------------------------------------------------------ other BASELINE configs
@pytest.mark.parametrize("cfg_name,storage", [("cfg5", "f32"), ("cfg5", "f16")])
def test_other_baseline_configs_match_oracle(cfg_name, storage):
    """BASELINE.json configs[4] (N=4, 640x512, D=192, interval 1.33, fp16) at full size against the CPU oracle
    with matched storage rounding, and the fp32-storage run of the same shape at fp32 tolerance.  configs[2]
    (N=5, 1600x1184 -> 296x400, D=256, bf16 and fp32) runs the same comparison -- plus the variance volume and the
    plain fp32 oracle -- in tests/test_gpu_fullsize.py::test_16bit_configs_per_stage_and_against_the_fp32_oracle."""
    c = synthetic.CONFIGS[cfg_name]
    N, h, w, D = c["nviews"], c["H"] // 4, c["W"] // 4, c["D"]
    feats = synthetic.random_features(N, 32, h, w, seed=21)
    proj = synthetic.cameras(N, h, w, yaw_deg=0.5)
    dv = synthetic.depth_values(D, interval_scale=c["interval_scale"])
    sd = synthetic.random_costreg_state(seed=2)
    depth, conf = hip_depth_infer(feats, proj, dv, sd, dtype=_lib.dtype_code(storage))
    depth_o, conf_o = orc.depth_infer(feats, proj, dv, sd, storage=storage)
    assert np.isfinite(depth).all()
    if storage == "f32":
        assert rel_l1(depth, depth_o) < 5e-6
        assert (np.abs(conf - conf_o) > 5e-3).mean() < 0.01
    else:
        assert rel_l1(depth, depth_o) < (2e-4 if storage == "f16" else 1e-3)


# ------------------------------------------------------------------------------ eval driver (f1)
def test_save_depth_sharded_writes_reference_file_tree(tmp_path):
    """Sharded save_depth counterpart: file tree, PFM payload == model output, cam text."""
    from scene_3dreconstruction_mvsnet_amd import data_io
    from scene_3dreconstruction_mvsnet_amd.eval_driver import save_depth_sharded
    w = load_weights()
    model = MVSNet(refine=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    samples = []
    for i in range(3):
        imgs, proj, dv = synthetic.make_inputs(3, 64, 96, 16, seed=40 + i)
        samples.append({"imgs": imgs[0], "proj_matrices": proj[0], "depth_values": dv[0],
                        "filename": "scan9/{}/" + "{:0>8}".format(i) + "{}",
                        "intrinsics": [np.eye(3, dtype=np.float32)] * 3,
                        "extrinsics": [np.eye(4, dtype=np.float32)] * 3})
    done = []
    for rank in range(2):  # two "ranks" run one after the other: together they cover every unit once
        done += save_depth_sharded(model, samples, str(tmp_path), rank=rank, world=2, device=DEV)
    assert sorted(done) == [0, 1, 2]
    model = model.to(DEV).eval()
    for i, s in enumerate(samples):
        d, _ = data_io.read_pfm(str(tmp_path / "scan9" / "depth_est" / f"{i:08d}.pfm"))
        c, _ = data_io.read_pfm(str(tmp_path / "scan9" / "confidence" / f"{i:08d}.pfm"))
        out = model(cu(s["imgs"][None]), cu(s["proj_matrices"][None]), cu(s["depth_values"][None]))
        np.testing.assert_array_equal(d, out["depth"][0].cpu().numpy())
        np.testing.assert_array_equal(c, out["photometric_confidence"][0].cpu().numpy())
        assert (tmp_path / "scan9" / "cams" / f"{i:08d}_cam.txt").read_text().startswith("extrinsic\n1.0 0.0 ")
        # reference image (eval.py:346-350) and the grey previews (eval.py:388,393)
        from PIL import Image
        img = np.array(Image.open(str(tmp_path / "scan9" / "images" / f"{i:08d}.png")))
        np.testing.assert_array_equal(img, np.uint8(np.transpose(s["imgs"][0], (1, 2, 0)) * 255))
        prev = np.array(Image.open(str(tmp_path / "scan9" / "depth_est" / f"{i:08d}.png")))
        np.testing.assert_array_equal(prev, np.uint8((d - d.min()) / (d.max() - d.min()) * 255))
        cprev = np.array(Image.open(str(tmp_path / "scan9" / "confidence" / f"{i:08d}.png")))
        assert cprev.shape == c.shape and set(np.unique(cprev)) <= {0, 1}


# ------------------------------------------------------------------------------ randomized shapes
@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_and_rigs_match_oracle(seed):
    """Seeded random problems: ragged tile edges in every axis, 1..9 views, rotated / widely spaced
    rigs (large out-of-image fractions), different depth ranges -- whole path vs the oracle."""
    rng = np.random.default_rng(100 + seed)
    N = int(rng.integers(1, 10))
    h, w, D = (int(8 * rng.integers(1, 6)), int(8 * rng.integers(1, 9)), int(8 * rng.integers(1, 5)))
    baseline = (float(rng.uniform(-120, 120)), float(rng.uniform(-40, 40)), float(rng.uniform(-5, 5)))
    feats = synthetic.random_features(N, 32, h, w, seed=seed)
    proj = synthetic.cameras(N, h, w, baseline=baseline, yaw_deg=float(rng.uniform(-3, 3)))
    dv = synthetic.depth_values(D, dmin=float(rng.uniform(300, 900)), interval=float(rng.uniform(1.0, 6.0)))
    sd = synthetic.random_costreg_state(seed=seed)
    depth, conf = hip_depth_infer(feats, proj, dv, sd)
    var = orc.variance_volume(feats, proj, dv)
    cost = orc.costreg_forward(var, sd)
    depth_o, conf_o, idx_o, prob_o = orc.softargmin_conf(cost, dv, want_prob=True)
    assert rel_l1(depth, depth_o) < 1e-5, (N, h, w, D)
    assert_conf_close(conf, conf_o, idx_o, prob=prob_o, atol=1e-3)


def test_dataset_to_pfm_end_to_end(tmp_path):
    """EvalDataset (f2) -> drop-in MVSNet -> sharded writer (f1) on a synthetic on-disk dataset."""
    import os
    from synthetic_dataset import write_synthetic_dataset
    from scene_3dreconstruction_mvsnet_amd import data_io
    from scene_3dreconstruction_mvsnet_amd.dataset_eval import EvalDataset
    from scene_3dreconstruction_mvsnet_amd.eval_driver import save_depth_sharded
    listfile = write_synthetic_dataset(str(tmp_path))
    ds = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, "test", 3, 16, 1.06,
                     img_res=(96, 128), dataset_name="dtu")
    w = load_weights()
    model = MVSNet(refine=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    out = tmp_path / "out"
    done = save_depth_sharded(model, ds, str(out), rank=1, world=4, device=DEV)
    assert done == [1, 5]
    d, _ = data_io.read_pfm(str(out / "scan1" / "depth_est" / "00000001.pfm"))
    assert d.shape == (24, 32) and np.isfinite(d).all()
    dv = ds[1]["depth_values"]
    assert d.min() >= dv[0] - 1e-3 and d.max() <= dv[-1] + 1e-3
    assert (out / "scan9" / "confidence" / "00000001.pfm").exists()
    # the same run fed by decoder PROCESSES through the shared-memory ring (decoder_pool.py): same files
    out2 = tmp_path / "out_procs"
    ds2 = EvalDataset(os.path.join(str(tmp_path), "data"), listfile, "test", 3, 16, 1.06,
                      img_res=(96, 128), dataset_name="dtu", cache_images=8)
    assert save_depth_sharded(model, ds2, str(out2), rank=1, world=4, device=DEV, decoder_procs=2) == [1, 5]   # view-level pool
    from scene_3dreconstruction_mvsnet_amd.decoder_pool import DecoderPool
    out3 = tmp_path / "out_procs_samples"
    with DecoderPool(ds2, procs=2, chunk=1) as pool:                                                                # sample-level pool
        assert save_depth_sharded(model, ds2, str(out3), rank=1, world=4, device=DEV, decoder_pool=pool) == [1, 5]
    assert (out / "scan1/depth_est/00000001.pfm").read_bytes() == (out3 / "scan1/depth_est/00000001.pfm").read_bytes()
    for rel in ("scan1/depth_est/00000001.pfm", "scan9/confidence/00000001.pfm", "scan1/images/00000001.png",
                "scan9/cams/00000001_cam.txt"):
        assert (out / rel).read_bytes() == (out2 / rel).read_bytes(), rel


# ------------------------------------------------------------------------------ replicas / threads (b5)
def test_forward_on_a_dataparallel_replica():
    """nn.DataParallel with more than one visible GPU runs forward on torch.nn.parallel.replicate()
    copies whose parameters are not in `_parameters`; weights must come from the source module."""
    fx = load_fixture("small")
    w = load_weights()
    model = MVSNet(refine=False).to(DEV).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    replica = torch.nn.parallel.replicate(model, [0])[0]
    assert "conv0.conv.weight" not in replica.cost_regularization.state_dict()
    args = (cu(fx["imgs"]), cu(fx["proj_matrices"]), cu(fx["depth_values"]))
    for impl in ("hip", "torch"):
        model.feature_impl = impl
        replica = torch.nn.parallel.replicate(model, [0])[0]
        out = replica(*args)
        assert rel_l1(out["depth"].cpu().numpy(), fx["depth"]) < 1e-4
        assert torch.equal(out["depth"], model(*args)["depth"])
    if torch.cuda.device_count() > 1:      # the default DataParallel over every visible device, B = 2
        fx2 = load_fixture("b2")
        dp = torch.nn.DataParallel(model)
        out = dp(cu(fx2["imgs"]), cu(fx2["proj_matrices"]), cu(fx2["depth_values"]))
        assert rel_l1(out["depth"].cpu().numpy(), fx2["depth"]) < 1e-4


def test_two_host_threads_on_two_streams_share_one_module():
    """Two host threads, each on its own HIP stream, drive the same module with the same shape:
    the per-(device, stream) workspaces keep their volumes apart."""
    import threading
    w = load_weights()
    model = MVSNet(refine=False).to(DEV).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    problems = []
    for seed in (1, 2):
        imgs, proj, dv = synthetic.make_inputs(3, 128, 160, 48, seed=seed)
        problems.append((cu(imgs), cu(proj), cu(dv)))
    want = [model(*p)["depth"].clone() for p in problems]
    torch.cuda.synchronize()
    got = [[], []]
    errors = []

    def worker(i):
        try:
            st = torch.cuda.Stream(DEV)
            with torch.cuda.stream(st):
                for _ in range(20):
                    got[i].append(model(*problems[i])["depth"])
            st.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for d in got[i]:
            assert torch.equal(d, want[i])
    keys = [k for k in model._workspace_cache if k[0] == "fwd"]
    assert len({k[2] for k in keys}) >= 3          # default stream + the two side streams


def test_save_depth_raises_when_the_completer_fails(tmp_path):
    """A failure while copying a finished map to the host / handing it to the writers must surface
    (it used to be lost with the thread: missing PFMs, or a hang on the full queue)."""
    from scene_3dreconstruction_mvsnet_amd.eval_driver import save_depth_sharded

    class Broken(torch.nn.Module):
        def forward(self, imgs, proj, dv):
            return {"depth": torch.zeros((1, 4, 4), device=imgs.device)}   # no photometric_confidence

    samples = []
    for i in range(70):   # more than the completion queue holds: the old code hung here
        samples.append({"imgs": np.zeros((2, 3, 16, 16), np.float32), "proj_matrices": np.zeros((2, 4, 4), np.float32),
                        "depth_values": np.arange(8, dtype=np.float32),
                        "filename": "scanX/{}/" + "{:0>8}".format(i) + "{}"})
    with pytest.raises(RuntimeError, match="incomplete"):
        save_depth_sharded(Broken(), samples, str(tmp_path), device=DEV, writers=1, decoders=2)


# ------------------------------------------------------------------------------ more than one GPU
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the round's test box has one)")
def test_two_ranks_gather_over_rccl_and_bench_self_launch():
    """On a multi-GPU box: `python bench.py --gpus 2` launches its own ranks (one per GPU, RCCL all-gather
    of the results inside the timed region) and prints one line for the whole job.  Skipped -- and so
    UNMEASURED -- on the one-GPU box these tests normally run on."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline", "--no-e2e", "--staged-steps", "0"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"


@pytest.mark.parametrize("N,h,w,D,kw", [
    (5, 32, 40, 48, {}),                                   # cfg1-like rig, bands of out-of-image samples
    (4, 24, 72, 32, dict(baseline=(-90.0, 25.0, 0.0))),    # most samples of the far views outside (whole waves skip)
    (3, 16, 24, 16, dict(baseline=(-1e5, 0.0, 0.0))),      # every source sample outside: variance of the reference alone
    (5, 40, 56, 24, dict(yaw_deg=2.5)),                    # rotated rig: cells change often, ragged last block
])
def test_tap_cache_kernel_is_bit_identical_to_the_plain_kernel(tmp_path, N, h, w, D, kw):
    """warp_variance_tc2 (taps cached across depth, zero-weight views skipped per wave, lean projection) against the
    plain gather kernel (MVS_WARP_TC=0, read once per process -> child process): same taps, same weights, same fma
    nesting, so the volumes must be EQUAL, not close."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    feats = synthetic.random_features(N, 32, h, w, seed=11)
    proj = synthetic.cameras(N, h, w, **kw)
    dv = synthetic.depth_values(D)
    np.savez(tmp_path / "in.npz", feats=feats, proj=proj, dv=dv)
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from scene_3dreconstruction_mvsnet_amd import _lib\n"
        "z = np.load(sys.argv[1]); dev = 'cuda:0'; cu = lambda a: torch.from_numpy(a).to(dev)\n"
        "N, C, h, w = z['feats'].shape\n"
        "ws = _lib.alloc_workspace(N, C, z['dv'].shape[0], h, w, dev)\n"
        "v = _lib.warp_variance(cu(z['feats']), _lib.relative_proj(cu(z['proj'])), cu(z['dv']), ws)\n"
        "np.save(sys.argv[2], v.cpu().numpy())\n") % os.path.dirname(here)
    r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "in.npz"), str(tmp_path / "plain.npy")],
                       env=dict(os.environ, MVS_WARP_TC="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    ws = _lib.alloc_workspace(N, 32, D, h, w, DEV)
    got = _lib.warp_variance(cu(feats), _lib.relative_proj(cu(proj)), cu(dv), ws).cpu().numpy()
    want = np.load(tmp_path / "plain.npy")
    assert np.isfinite(got).all()
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    ref = orc.variance_volume(feats, proj, dv)
    np.testing.assert_allclose(_lib.from_c8(torch.from_numpy(got)).numpy(), ref, rtol=0, atol=5e-4)
