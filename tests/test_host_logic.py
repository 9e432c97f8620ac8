"""CPU-side checks: the C-ABI library loads and exports every declared symbol, argument
validation / error strings, weight packing (BN fold + layout) and the drop-in module's
checkpoint-key compatibility.  No kernel is launched here (no GPU in this suite)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO, load_weights
from scene_3dreconstruction_mvsnet_amd import MVSNet, _lib


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "mvs_abi.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(mvs_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert _lib.load().mvs_abi_version() == 2


def test_query_workspace_and_shape_errors():
    n = _lib.query_workspace(5, 32, 192, 128, 160)
    # variance volume alone is 503 MB at cfg2
    assert n > 192 * 128 * 160 * 32 * 4
    assert n % 256 == 0
    for bad in [(5, 16, 192, 128, 160), (5, 32, 190, 128, 160), (5, 32, 192, 130, 160),
                (5, 32, 192, 128, 164), (0, 32, 192, 128, 160), (5, 32, 0, 128, 160)]:
        with pytest.raises(_lib.MvsError) as e:
            _lib.query_workspace(*bad)
        assert e.value.code == 1, bad  # MVS_ERR_BAD_SHAPE
    with pytest.raises(_lib.MvsError) as e:
        _lib.query_workspace(5, 32, 192, 128, 160, dtype=7)
    assert e.value.code == 2  # MVS_ERR_BAD_DTYPE
    assert "dtype" in str(e.value)


def test_null_pointer_is_an_error_not_a_crash():
    lib = _lib.load()
    assert lib.mvs_query_workspace(5, 32, 192, 128, 160, 0, None) == 5  # MVS_ERR_NULL
    assert lib.mvs_depth_infer(None, None, None, None, None, None, None, 0,
                               5, 32, 192, 128, 160, 0, None) == 5
    assert b"NULL" in lib.mvs_last_error_string()


def _costreg_state(weights):
    return {k[len("cost_regularization."):]: v for k, v in weights.items()
            if k.startswith("cost_regularization.")}


def test_pack_weights_folds_bn_and_relayouts():
    w = load_weights()
    st = _costreg_state(w)
    blob = _lib.pack_weights(st).numpy().view(np.float32)
    # recompute the layout in numpy: sections are 64-float aligned, [27][cin][cout] then bias
    off = 0
    layer_ch = _lib._LAYER_CH
    for l, key in enumerate(_lib.CONV_WEIGHT_KEYS):
        ci, co = layer_ch[l]
        wt = st[key].astype(np.float64)
        if 7 <= l <= 9:
            wt = wt.transpose(1, 0, 2, 3, 4)  # [Cin][Cout] -> [Cout][Cin]
        if l < 10:
            pre = _lib.BN_PREFIXES[l]
            scale = st[pre + ".weight"] / np.sqrt(st[pre + ".running_var"].astype(np.float64) + 1e-5)
            shift = st[pre + ".bias"] - st[pre + ".running_mean"] * scale
        else:
            scale = np.ones(co)
            shift = st["prob.bias"].astype(np.float64)
        want = (wt * scale[:, None, None, None, None]).reshape(co, ci, 27).transpose(2, 1, 0)
        got = blob[off:off + 27 * ci * co].reshape(27, ci, co)
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)
        off = (off + 27 * ci * co + 63) // 64 * 64
        np.testing.assert_allclose(blob[off:off + co], shift, rtol=2e-6, atol=1e-7)
        off = (off + co + 63) // 64 * 64
    # conv0 4x4x1 panel [4 chunks][27 taps][2 halves][2 nt][4 j][4 k]
    q = blob[off:off + 4 * 27 * 2 * 2 * 4 * 4].reshape(4, 27, 2, 2, 4, 4)
    w0t = blob[:27 * 32 * 8].reshape(27, 32, 8)
    for c, tap, half, nt, j, k in [(0, 0, 0, 0, 0, 0), (3, 26, 1, 1, 3, 3), (1, 13, 0, 1, 2, 1)]:
        assert q[c, tap, half, nt, j, k] == w0t[tap, 8 * c + 4 * half + k, 4 * nt + j]
    off += 4 * 27 * 2 * 2 * 4 * 4
    # generic MFMA panels of layers 1..6: [cin/8][cout/16][14 k-steps][64 lanes][4]
    woffs, o = [], 0
    for l in range(11):
        ci, co = layer_ch[l]
        woffs.append(o)
        o = (o + 27 * ci * co + 63) // 64 * 64
        o = (o + co + 63) // 64 * 64
    for l in range(1, 7):
        ci, co = layer_ch[l]
        nch, nt = ci // 8, co // 16
        panel = blob[off:off + nch * nt * 14 * 64 * 4].reshape(nch, nt, 14, 64, 4)
        wl = blob[woffs[l]:woffs[l] + 27 * ci * co].reshape(27, ci, co)
        for c, t, ks, lane in [(0, 0, 0, 0), (nch - 1, nt - 1, 13, 63), (nch // 2, 0, 7, 21)]:
            g, n = lane >> 4, lane & 15
            tap = 2 * ks + (g >> 1)
            for j4 in range(4):
                want = wl[tap, 8 * c + 4 * (g & 1) + j4, 16 * t + n] if tap < 27 else 0.0
                assert panel[c, t, ks, lane, j4] == want
        off += nch * nt * 14 * 64 * 4
    for l in range(7, 10):  # deconv panels [cin/8][2*cout/16][9][64][4]
        ci, co = layer_ch[l]
        off += (ci // 8) * (2 * co // 16) * 9 * 64 * 4
    # 16-bit MFMA panels (fp16 then bf16), layers 0..9; spot-check the conv2 panel's rounding
    for d, npdt in enumerate(("f16", "bf16")):
        for l in range(10):
            ci, co = layer_ch[l]
            elems = 4 * 9 * 64 * 8 if l == 0 else (ci // 8) * (co // 16) * 7 * 64 * 8 if l <= 6 \
                else (ci // 8) * (2 * co // 16) * 5 * 64 * 8
            if l == 9 and npdt == "bf16":
                h16_bf16_off9 = off
            if l == 2:
                panel = blob[off:off + elems // 2].view(np.uint16).reshape(ci // 8, co // 16, 7, 64, 8)
                wl = blob[woffs[l]:woffs[l] + 27 * ci * co].reshape(27, ci, co)
                for c, t, ks, lane, j in [(0, 0, 0, 0, 0), (1, 0, 6, 63, 7), (1, 0, 3, 37, 2)]:
                    g, n = lane >> 4, lane & 15
                    tap = 4 * ks + g
                    want = wl[tap, 8 * c + j, 16 * t + n] if tap < 27 else np.float32(0)
                    if npdt == "f16":
                        bits = np.array([want], np.float32).astype(np.float16).view(np.uint16)[0]
                    else:
                        u = int(np.array([want], np.float32).view(np.uint32)[0])
                        bits = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
                    assert panel[c, t, ks, lane, j] == bits
            off += (elems // 2 + 63) // 64 * 64
    # Winograd-z panels of the stride-1 layers 2 and 4: [cin/8][4 t][cout/16][5 k-steps][64 lanes][4]
    for l in (2, 4):
        ci, co = layer_ch[l]
        nch, nt = ci // 8, co // 16
        panel = blob[off:off + nch * 4 * nt * 5 * 256].reshape(nch, 4, nt, 5, 64, 4)
        gl = blob[woffs[l]:woffs[l] + 27 * ci * co].reshape(3, 9, ci, co).astype(np.float64)
        Gl = [gl[0], (gl[0] + gl[1] + gl[2]) / 2, (gl[0] - gl[1] + gl[2]) / 2, gl[2]]
        for c, t, n, ks, lane, j4 in [(0, 0, 0, 0, 0, 0), (nch - 1, 3, nt - 1, 4, 63, 3), (nch // 2, 1, 0, 2, 21, 1),
                                      (0, 2, nt - 1, 4, 5, 2)]:
            g, col = lane >> 4, lane & 15
            tap = 2 * ks + (g >> 1)
            want = Gl[t][tap, 8 * c + 4 * (g & 1) + j4, 16 * n + col] if tap < 9 else 0.0
            np.testing.assert_allclose(panel[c, t, n, ks, lane, j4], want, rtol=1e-6, atol=1e-9)
        off += nch * 4 * nt * 5 * 256
    # conv0 Winograd F(4,3)-z panel [4 chunks][6 t][9 taps][2 halves][2 nt][4 j][4 k] (conv_winograd.hip)
    w43 = blob[off:off + 4 * 6 * 9 * 2 * 2 * 4 * 4].reshape(4, 6, 9, 2, 2, 4, 4)
    g = blob[:27 * 32 * 8].reshape(3, 9, 32, 8).astype(np.float64)
    G43 = [g[0] / 4, -(g[0] + g[1] + g[2]) / 6, -(g[0] - g[1] + g[2]) / 6,
           g[0] / 24 + g[1] / 12 + g[2] / 6, g[0] / 24 - g[1] / 12 + g[2] / 6, g[2]]
    for c, t, tap, half, nt, j, k in [(0, 0, 0, 0, 0, 0, 0), (3, 5, 8, 1, 1, 3, 3), (1, 1, 4, 0, 1, 2, 1),
                                      (2, 3, 5, 1, 0, 1, 3), (0, 4, 7, 0, 0, 3, 2), (3, 2, 2, 1, 1, 0, 0)]:
        np.testing.assert_allclose(w43[c, t, tap, half, nt, j, k], G43[t][tap, 8 * c + 4 * half + k, 4 * nt + j],
                                   rtol=1e-6, atol=1e-9)
    off += 4 * 6 * 9 * 2 * 2 * 4 * 4
    # the same transformed weights as three bf16 pieces, Toeplitz-pair panel [4][6 t][3 ky][3 pieces][64 lanes][8]
    # (conv0_split.hip): the pieces of every weight add up to the fp32 value to <= 2^-24 of it; columns with
    # kx = g - jj outside [0, 2] are zero
    n16 = 4 * 6 * 3 * 3 * 64 * 8
    ps = blob[off:off + n16 // 2].view(np.uint16).reshape(4, 6, 3, 3, 64, 8)
    pf = (ps.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    for c, t, ky, lane, j in [(0, 0, 0, 0, 0), (3, 5, 2, 63, 7), (1, 1, 1, 21, 3), (2, 3, 0, 40, 5), (0, 4, 2, 9, 1),
                              (3, 2, 1, 55, 6), (1, 0, 0, 8, 0), (2, 5, 1, 31, 4)]:
        n, gq = lane & 15, lane >> 4
        jj, co, kx = n >> 3, n & 7, (lane >> 4) - ((lane & 15) >> 3)
        want = float(np.float32(G43[t][ky * 3 + kx, 8 * c + j, co])) if 0 <= kx <= 2 else 0.0
        got = pf[c, t, ky, :, lane, j]
        assert abs(got.sum() - want) <= 2.0 ** -24 * abs(want), (c, t, ky, lane, j, got, want)
        if want != 0.0:
            assert abs(got[1]) <= 2.0 ** -8 * abs(want) and abs(got[2]) <= 2.0 ** -16 * abs(want)
    off += n16 // 2
    # split-operand panels of layers 2..4: three bf16 pieces, each in the layout of the layer's bf16 panel (h16);
    # piece 0 IS the bf16 panel (RNE of the folded weights), and the three pieces add up to the fp32 panel value
    for l in range(2, 5):
        ci, co = _lib._LAYER_CH[l]
        elems = (ci // 8) * (co // 16) * 7 * 64 * 8
        pcs = blob[off:off + (3 * elems) // 2].view(np.uint16).reshape(3, elems)
        f = (pcs.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        assert np.all(np.abs(f[1]) <= 2.0 ** -8 * np.abs(f[0]) + 1e-30) and np.all(np.abs(f[2]) <= 2.0 ** -16 * np.abs(f[0]) + 1e-30)
        assert (f[0] != 0).mean() > 0.5
        off += ((3 * elems) // 2 + 63) // 64 * 64
    # the same for conv11 (layer 9, the fused tail's transposed convolution): [3 pieces][2 chunks][5 k-steps][64][8];
    # piece 0 is the layer's bf16 panel of the 16-bit modes
    elems = 2 * 1 * 5 * 64 * 8
    pcs = blob[off:off + (3 * elems) // 2].view(np.uint16).reshape(3, elems)
    f = (pcs.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.all(np.abs(f[1]) <= 2.0 ** -8 * np.abs(f[0]) + 1e-30) and np.all(np.abs(f[2]) <= 2.0 ** -16 * np.abs(f[0]) + 1e-30)
    assert (f[0] != 0).mean() > 0.5
    h9 = blob[h16_bf16_off9:h16_bf16_off9 + elems // 2].view(np.uint16)
    assert np.array_equal(h9, pcs[0])
    off += ((3 * elems) // 2 + 63) // 64 * 64
    # ... conv9 (layer 8) and conv7 (layer 7), the transposed layers' split-operand kernels (deconvgs)
    for l in (8, 7):
        ci, co = _lib._LAYER_CH[l]
        elems = (ci // 8) * (2 * co // 16) * 5 * 64 * 8
        pcs = blob[off:off + (3 * elems) // 2].view(np.uint16).reshape(3, elems)
        f = (pcs.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        assert np.all(np.abs(f[1]) <= 2.0 ** -8 * np.abs(f[0]) + 1e-30) and np.all(np.abs(f[2]) <= 2.0 ** -16 * np.abs(f[0]) + 1e-30)
        assert (f[0] != 0).mean() > 0.5
        off += ((3 * elems) // 2 + 63) // 64 * 64
    assert off * 4 == _lib.query_weights_blob()


def test_pack_weights_rejects_wrong_shapes():
    st = dict(_costreg_state(load_weights()))
    st["conv0.conv.weight"] = st["conv0.conv.weight"][:, :16]
    with pytest.raises(RuntimeError):
        _lib.pack_weights(st)


def test_state_dict_keys_match_reference_checkpoint():
    w = load_weights()  # keys captured from the reference's MVSNet(refine=False).state_dict()
    m = MVSNet(refine=False)
    assert set(m.state_dict().keys()) == set(w.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(w[k].shape), k
    # eval.py:309-315: checkpoint keys carry the nn.DataParallel "module." prefix
    ckpt = {"module." + k: torch.from_numpy(v) for k, v in w.items()}
    dp = torch.nn.DataParallel(MVSNet(refine=False))
    missing = dp.load_state_dict(ckpt, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert 338_129 == sum(p.numel() for p in m.parameters())


def test_forward_refuses_cpu_training_and_view_mismatch():
    m = MVSNet(refine=False).eval()
    imgs = torch.zeros(1, 3, 3, 32, 32)
    proj = torch.eye(4).repeat(1, 3, 1, 1)
    dv = torch.linspace(425, 500, 8)[None]
    with pytest.raises(RuntimeError, match="no CPU"):
        m(imgs, proj, dv)
    with pytest.raises(AssertionError, match="Different number"):
        m(imgs, proj[:, :2], dv)
    with pytest.raises(RuntimeError, match="eval"):
        m.train()(imgs, proj, dv)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmvs_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_module_survives_deepcopy_and_pickle():
    import copy
    import io
    m = MVSNet(refine=False)
    m2 = copy.deepcopy(m)
    assert set(m2.state_dict()) == set(m.state_dict()) and m2._blob_cache == {}
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert isinstance(m3, MVSNet) and m3._workspace_cache == {}


def test_pack_feature_weights_folds_bn_and_relayouts():
    """FeatureNet blob: per layer the MFMA panel [cin/8][cout/16][(k*k+1)/2][64][4] of the BN-folded
    weights, then the folded bias; conv0 additionally as a plain [27][8] table (fused kernel)."""
    w = load_weights()
    st = {k[len("feature."):]: v for k, v in w.items() if k.startswith("feature.")}
    blob = _lib.pack_feature_weights(st).numpy().view(np.float32)
    off = 0
    folded0 = None
    for l, (ci, co, k, _) in enumerate(_lib.FEATURE_LAYERS):
        wt = st[_lib.FEATURE_WEIGHT_KEYS[l]].astype(np.float64)
        if l < 7:
            scale = st[f"conv{l}.bn.weight"] / np.sqrt(st[f"conv{l}.bn.running_var"].astype(np.float64) + 1e-5)
            shift = st[f"conv{l}.bn.bias"] - st[f"conv{l}.bn.running_mean"] * scale
        else:
            scale, shift = np.ones(co), st["feature.bias"].astype(np.float64)
        wf = wt * scale[:, None, None, None]
        if l == 0:
            folded0 = (wf, shift)
        nch, nt, ks = (ci + 7) // 8, (co + 15) // 16, (k * k + 1) // 2
        panel = blob[off:off + nch * nt * ks * 256].reshape(nch, nt, ks, 64, 4)
        rng = np.random.default_rng(l)
        for _ in range(40):
            c, t, s, lane, j4 = rng.integers(nch), rng.integers(nt), rng.integers(ks), rng.integers(64), rng.integers(4)
            g, n = lane >> 4, lane & 15
            tap, cin_i, cout_i = 2 * s + (g >> 1), 8 * c + 4 * (g & 1) + j4, 16 * t + n
            want = wf[cout_i, cin_i, tap // k, tap % k] if (tap < k * k and cin_i < ci and cout_i < co) else 0.0
            np.testing.assert_allclose(panel[c, t, s, lane, j4], want, rtol=2e-6, atol=1e-8)
        off += nch * nt * ks * 256
        np.testing.assert_allclose(blob[off:off + co], shift, rtol=2e-6, atol=1e-7)
        off += (nt * 16 + 63) // 64 * 64
    direct = blob[off:off + 28 * 8].reshape(28, 8)
    np.testing.assert_allclose(direct[:27], folded0[0].reshape(8, 27).T, rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(direct[27], folded0[1], rtol=2e-6, atol=1e-7)
    assert (off + 256) * 4 == _lib.query_feature_blob()
    bad = dict(st)
    bad["conv2.conv.weight"] = bad["conv2.conv.weight"][:, :, :3, :3]
    with pytest.raises(RuntimeError):
        _lib.pack_feature_weights(bad)


def test_feature_and_forward_workspace_queries():
    full = 5 * 512 * 640 * 8 * 4
    assert _lib.query_feature_workspace(5, 512, 640) == 2 * full + 5 * 32 * 128 * 160 * 4
    assert _lib.query_forward_workspace(5, 512, 640, 192) == \
        _lib.query_workspace(5, 32, 192, 128, 160) + _lib.query_feature_workspace(5, 512, 640)
    for bad in [(5, 510, 640, 192), (5, 512, 640, 190), (0, 512, 640, 192)]:
        with pytest.raises(_lib.MvsError) as e:
            _lib.query_forward_workspace(*bad)
        assert e.value.code == 1
    with pytest.raises(_lib.MvsError):
        _lib.query_feature_workspace(1, 2, 2)
    lib = _lib.load()
    assert lib.mvs_forward_images(None, None, None, None, None, None, None, None, 0, 5, 512, 640, 192, 0, None) == 5
    assert lib.mvs_feature_net(None, None, None, None, 0, 5, 512, 640, None) == 5


# ---------------------------------------------------------------------- nn.DataParallel replicas
def _emulated_replica(model):
    """What torch.nn.parallel.replicate() builds for one device, without a GPU: every module is
    copied with `_replicate_for_data_parallel()`, children are re-linked to the copies, and the
    parameters become PLAIN attributes holding broadcast copies (not `_parameters` entries)."""
    modules = list(model.modules())
    index = {m: i for i, m in enumerate(modules)}
    copies = [m._replicate_for_data_parallel() for m in modules]
    for m, c in zip(modules, copies):
        for name, child in m._modules.items():
            c._modules[name] = None if child is None else copies[index[child]]
        for name, p in m._parameters.items():
            if p is not None:
                setattr(c, name, p.detach().clone())     # new storage every forward, as broadcast does
        for name, b in m._buffers.items():
            if b is not None:
                c._buffers[name] = b.detach().clone()
    return copies[0]


def test_weights_are_packed_from_the_source_module_on_a_dataparallel_replica():
    """eval.py:309 wraps the model in nn.DataParallel; with >1 visible GPU forward runs on replicas
    whose parameters are not in `_parameters` -- state_dict() on them has no conv weights."""
    w = load_weights()
    model = MVSNet(refine=False).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    rep = _emulated_replica(model)
    assert "conv0.conv.weight" not in rep.cost_regularization.state_dict()   # the trap
    assert rep._source() is model and _emulated_replica(rep)._source() is model
    for which in ("cost_regularization", "feature"):
        want = getattr(model, which).state_dict()
        got = rep._packable_state(which)
        assert set(got) == set(want)
        for k in want:
            assert torch.equal(got[k], want[k]), k
    blob_src = _lib.pack_weights(model._packable_state("cost_regularization"))
    blob_rep = _lib.pack_weights(rep._packable_state("cost_regularization"))
    assert torch.equal(blob_src, blob_rep)
    # the blob cache key follows the SOURCE parameters: stable across replicas, moves with updates
    v0 = MVSNet._param_versions(rep._source().cost_regularization)
    assert v0 == MVSNet._param_versions(_emulated_replica(model)._source().cost_regularization)
    with torch.no_grad():
        model.cost_regularization.prob.weight.mul_(2.0)
    assert v0 != MVSNet._param_versions(rep._source().cost_regularization)
    # caches are shared with the source, the back-reference is not pickled
    assert rep._blob_cache is model._blob_cache and rep._cache_lock is model._cache_lock
    assert "_dp_source" not in rep.__getstate__()


def test_workspace_cache_is_bounded_and_most_recently_used_wins():
    model = MVSNet(refine=False)
    made = []

    def alloc(key):
        return model._cached_workspace(key, lambda: made.append(key) or 16, torch.device("cpu"))

    first = alloc("a")
    assert alloc("a") is first and made == ["a"]
    for k in range(model._MAX_CACHED_WORKSPACES - 1):
        alloc(("k", k))
    assert alloc("a") is first                       # still cached, now most recently used
    alloc("overflow")                                # evicts the oldest (("k", 0)), not "a"
    assert len(model._workspace_cache) == model._MAX_CACHED_WORKSPACES
    assert "a" in model._workspace_cache and ("k", 0) not in model._workspace_cache


# ---------------------------------------------------------------------- bench.py plumbing
def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_self_launch_command_and_defaults():
    bench = _bench()
    args = bench.parse_args([])
    assert args.gpus == 1 and args.steps > 0 and args.warmup >= 0 and args.config == "cfg2"
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    cmd = bench.self_launch_command(bench.parse_args(argv), argv, port=29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == argv                       # the ranks see the caller's flags unchanged
    assert 1024 < bench.free_port() < 65536
    assert set(bench.CONFIG_NAMES) == set(__import__(
        "scene_3dreconstruction_mvsnet_amd.synthetic", fromlist=["CONFIGS"]).CONFIGS)


def test_bench_path_totals_equal_survey_d3():
    """algorithmic bytes / FLOPs per map, layer by layer, no fusion credited -- whichever
    kernels ran (conv11 + prob count once, not again as the fused `conv11_prob` stage)."""
    bench = _bench()
    want = {  # (N, D, h, w, storage): (GB, GFLOP) as SURVEY §8 d3 prints them
        (3, 48, 32, 40, "f32"): (0.0310, 1.25), (5, 192, 128, 160, "f32"): (1.9636, 79.84),
        (5, 256, 296, 400, "bf16"): (7.556, 615.4), (4, 192, 128, 160, "f16"): (0.9806, 79.84)}
    for (N, D, h, w, st), (gb, gf) in want.items():
        costs = bench.stage_costs(N, D, h, w, 4 if st == "f32" else 2)
        b, f, floor_s = bench.path_totals(costs, bench.mfma_peak_tflops(st))
        assert abs(b / 1e9 - gb) < 6e-4 * max(1, gb), (st, b)
        assert abs(f / 1e9 - gf) < 6e-3 * max(1, gf / 10), (st, f)
        if st != "f32":   # priced against the 16-bit matrix peak the big stages are HBM-bound (conv0: 3.5x)
            assert costs["conv0"]["bytes"] / 8e12 > costs["conv0"]["flops"] / (bench.mfma_peak_tflops(st) * 1e12)
    costs = bench.stage_costs(5, 192, 128, 160, 4)
    b, f, floor_s = bench.path_totals(costs, bench.mfma_peak_tflops("f32"))
    assert b == 1_963_622_400 and abs(floor_s * 1e3 - 0.5948) < 1e-4
    assert bench.mfma_peak_tflops("f16", mfma16=False) == bench.MFMA_F32_PEAK_TFLOPS


def test_bench_parent_of_a_multi_gpu_run_never_loads_torch(tmp_path):
    """`python bench.py --gpus 2` without torchrun: the parent only relays; it must not initialise
    HIP (it must not even import torch) before starting the ranks."""
    import subprocess
    import sys
    fake = tmp_path / "fake_torchrun.py"
    fake.write_text("import json, sys\nprint('noise line')\n"
                    "print(json.dumps({'metric': 'm', 'n_gpus': 2, 'argv': sys.argv[1:]}))\n")
    code = (
        "import sys, json; sys.argv = ['bench.py', '--gpus', '2', '--steps', '3']\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "import importlib.util\n"
        f"spec = importlib.util.spec_from_file_location('bench_mod', {os.path.join(REPO, 'bench.py')!r})\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        f"b.self_launch_command = lambda args, argv, port=None: [sys.executable, {str(fake)!r}] + list(argv)\n"
        "try:\n    b.main()\nexcept SystemExit as e:\n    rc = e.code\n"
        "assert 'torch' not in sys.modules, 'parent imported torch'\n"
        "sys.exit(rc)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout              # exactly ONE JSON line on stdout
    import json
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["argv"] == ["--gpus", "2", "--steps", "3"]
    assert "noise line" in r.stderr


def test_in_image_fraction():
    from scene_3dreconstruction_mvsnet_amd import synthetic
    h, w = 32, 40
    dv = synthetic.depth_values(16)
    same = np.stack([synthetic.cameras(1, h, w)[0]] * 3)          # identical cameras: every point lands
    assert synthetic.in_image_fraction(same, dv, h, w) == 1.0
    assert synthetic.in_image_fraction(same[:1], dv, h, w) == 1.0  # no source view
    far = synthetic.cameras(2, h, w, baseline=(-1e5, 0.0, 0.0))    # source looks somewhere else entirely
    assert synthetic.in_image_fraction(far, dv, h, w) == 0.0
    f = synthetic.in_image_fraction(synthetic.cameras(5, h, w), dv, h, w)
    assert 0.5 < f < 1.0


# ---------------------------------------------------------------------- rank -> host cores
def test_rank_cpus_follow_the_gpu_numa_node(tmp_path):
    from scene_3dreconstruction_mvsnet_amd import sharding
    assert sharding._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    # fake sysfs: 4 GPUs, two per socket, plus a NIC and an AMD non-GPU function that must be skipped
    devs = {"0000:05:00.0": ("0x1002", "0x038000", "0-7,16-23"), "0000:15:00.0": ("0x1002", "0x038000", "0-7,16-23"),
            "0000:85:00.0": ("0x1002", "0x120000", "8-15,24-31"), "0000:95:00.0": ("0x1002", "0x120000", "8-15,24-31"),
            "0000:01:00.0": ("0x15b3", "0x020000", "0-7"), "0000:02:00.0": ("0x1002", "0x060400", "0-7")}
    for name, (vendor, cls, cpus) in devs.items():
        d = tmp_path / "bus/pci/devices" / name
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "class").write_text(cls + "\n")
        (d / "local_cpulist").write_text(cpus + "\n")
    gpus = sharding.gpu_local_cpus(str(tmp_path))
    assert len(gpus) == 4 and gpus[0] == list(range(8)) + list(range(16, 24))
    allowed = list(range(32))
    got = [sharding.rank_cpus(r, 4, allowed, gpus) for r in range(4)]
    assert got[0] == [0, 1, 2, 3, 4, 5, 6, 7] and got[1] == [16, 17, 18, 19, 20, 21, 22, 23]
    assert set(got[2]) | set(got[3]) == set(range(8, 16)) | set(range(24, 32))
    assert all(not (set(a) & set(b)) for i, a in enumerate(got) for b in got[i + 1:])   # disjoint
    # unknown topology (or a cgroup that hides the GPU's cores): contiguous slices of what is allowed
    assert sharding.rank_cpus(1, 2, [4, 5, 6, 7], None) == [6, 7]
    assert sharding.rank_cpus(0, 4, [40, 41, 42, 43], gpus) == [40]
    assert sharding.rank_cpus(3, 8, [0, 1], None) == [0, 1]       # fewer cores than ranks: share them
    assert sharding.pin_rank(0, 1) == []                          # a single rank is never pinned
    # the launcher may narrow / permute the visible devices: LOCAL_RANK counts the VISIBLE ones
    assert sharding.visible_gpu_order(4, {}) == [0, 1, 2, 3]
    assert sharding.visible_gpu_order(4, {"HIP_VISIBLE_DEVICES": "2,3"}) == [2, 3]
    assert sharding.visible_gpu_order(8, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "1,0"}) == [5, 4]
    assert sharding.visible_gpu_order(4, {"HIP_VISIBLE_DEVICES": "GPU-abc"}) == [0, 1, 2, 3]
    assert sharding.visible_gpu_order(2, {"CUDA_VISIBLE_DEVICES": "0,5"}) == [0, 1]     # out of range: ignored
    # CUDA_VISIBLE_DEVICES is an alias the HIP runtime reads only when HIP_VISIBLE_DEVICES is unset: both set to
    # the same permutation is ONE re-indexing
    assert sharding.visible_gpu_order(2, {"HIP_VISIBLE_DEVICES": "1,0", "CUDA_VISIBLE_DEVICES": "1,0"}) == [1, 0]
    assert sharding.visible_gpu_order(4, {"CUDA_VISIBLE_DEVICES": "3,1"}) == [3, 1]
    assert sharding.visible_gpu_order(4, {"HIP_VISIBLE_DEVICES": "2", "CUDA_VISIBLE_DEVICES": "3,1"}) == [2]
    # HIP's bus ids against the sorted-sysfs assumption
    cpus = [[0, 1], [2, 3], [4, 5]]
    assert sharding.check_gpu_order(cpus, ["0000:05:00.0", "0000:15:00.0", "0000:25:00.0"],
                                    ["0000:25:00.0", "0000:05:00.0", "0000:15:00.0"]) == [[4, 5], [0, 1], [2, 3]]
    assert sharding.check_gpu_order(cpus, ["0000:05:00.0"], ["0000:99:00.0"]) == cpus


def test_bench_roofline_fractions_never_exceed_one():
    """VERDICT r3 #2 / ADVICE: `frac` is computed from what the build EXECUTES (Winograd layers issue fewer
    multiply-adds, 16-bit modes keep fp32 logits), so no measured time can push it above 1 unless the kernel beats
    the hardware peak; the algorithmic d3 ratio travels separately and may exceed 1."""
    bench = _bench()
    N, D, h, w = 5, 192, 128, 160
    costs = bench.stage_costs(N, D, h, w, 4)
    exs = bench.executed_costs(costs, "f32", N, D, h, w, env={})      # default: conv0 with split bf16 operands
    assert exs["conv0"]["flops"] == costs["conv0"]["flops"] * 4 and exs["conv0"]["mfma_peak"] == bench.MFMA_16BIT_PEAK_TFLOPS
    ents = bench.stage_entry(0.276, costs["conv0"], exs["conv0"], bench.mfma_peak_tflops("f32"))
    assert ents["bound"] == "mfma" and 0.25 < ents["frac"] < 0.40 and ents["frac_algorithmic"] > 1.0 and "split" in ents["arith"]
    rs = bench.roofline_entry("conv0", 0.276, costs["conv0"], exs["conv0"], bench.mfma_peak_tflops("f32"))
    assert rs["peak"] == bench.MFMA_16BIT_PEAK_TFLOPS and rs["frac"] <= 1.0 and rs["algorithmic_ratio"] > 1.0
    ex = bench.executed_costs(costs, "f32", N, D, h, w, env={"MVS_CONV0_SPLIT": "0"})   # the fp32-MFMA Winograd kernel
    assert ex["conv0"]["flops"] == costs["conv0"]["flops"] * 0.5
    assert exs["conv2"]["mfma_peak"] == bench.MFMA_16BIT_PEAK_TFLOPS and abs(exs["conv2"]["flops"] / costs["conv2"]["flops"] - 6 * 28 / 27) < 1e-12
    ex = bench.executed_costs(costs, "f32", N, D, h, w, env={"MVS_CONV0_SPLIT": "0", "MVS_SPLIT_LAYERS": "0"})
    assert abs(ex["conv2"]["flops"] / costs["conv2"]["flops"] - 20 / 27) < 1e-12
    assert ex["conv1"] == costs["conv1"] and ex["conv0"]["bytes"] == costs["conv0"]["bytes"]
    assert exs["conv1"] == costs["conv1"]      # conv1 stays on the fp32-MFMA z-marching kernel
    assert bench.executed_costs(costs, "f32", N, D, h, w, env={"MVS_CONV0_WINO": "0"})["conv0"]["flops"] == costs["conv0"]["flops"]
    # the fused tail: conv11 with split operands on the bf16 matrix cores beside the prob stencil on the vector units --
    # its floor is the bytes (173 MB), not the sum of the two compute times; MVS_TAIL_SPLIT=0: all on the fp32 units
    et = bench.stage_entry(0.092, costs["conv11_prob"], exs["conv11_prob"], bench.mfma_peak_tflops("f32"))
    assert "parts" in exs["conv11_prob"] and et["bound"] == "hbm" and 0.2 < et["frac"] < 0.3 and "split" in et["arith"]
    e0 = bench.executed_costs(costs, "f32", N, D, h, w, env={"MVS_TAIL_SPLIT": "0"})["conv11_prob"]
    assert e0 == costs["conv11_prob"]
    et0 = bench.stage_entry(0.103, costs["conv11_prob"], e0, bench.mfma_peak_tflops("f32"))
    assert et0["bound"] == "mfma" and 0.25 < et0["frac"] < 0.4
    # round 3's driver line: conv0 0.3441 ms -> algorithmic 1.0043 of the fp32 MFMA peak, executed 0.502
    peak = bench.mfma_peak_tflops("f32")
    ent = bench.stage_entry(0.3441, costs["conv0"], ex["conv0"], peak)
    assert ent["bound"] == "mfma" and abs(ent["frac"] - 0.502) < 2e-3 and ent["frac_algorithmic"] > 1.0
    r = bench.roofline_entry("conv0", 0.3441, costs["conv0"], ex["conv0"], peak)
    assert r["frac"] <= 1.0 and abs(r["frac"] - 0.502) < 2e-3 and abs(r["algorithmic_ratio"] - 1.0043) < 2e-3
    assert r["executed_flops"] * 2 == r["algorithmic_flops"] and r["unit"] == "TFLOP/s"
    # 16-bit modes: fp32 logits are priced as moved (more than d3's algorithmic bytes)
    c16 = bench.stage_costs(5, 256, 296, 400, 2)
    e16 = bench.executed_costs(c16, "bf16", 5, 256, 296, 400, env={})
    V0 = 256 * 296 * 400
    assert e16["softargmin"]["bytes"] == c16["softargmin"]["bytes"] + 2 * V0
    assert e16["conv11_prob"]["bytes"] == c16["conv11_prob"]["bytes"] + 2 * V0
    assert e16["conv0"] == c16["conv0"]
    assert e16["warp_variance"]["bytes"] == c16["warp_variance"]["bytes"] + 5 * 32 * 296 * 400 * 2   # fp32 feature gather
    rw = bench.roofline_entry("warp_variance", 1.3, c16["warp_variance"], e16["warp_variance"], bench.mfma_peak_tflops("bf16"))
    assert rw["bound"] == "hbm" and rw["frac"] < 0.25


def test_bench_committed_traffic_follows_the_config():
    """cfg2 reads profiles/rNN_traffic.json, the other configs profiles/rNN_traffic_<cfg>.json; the warp kernel's
    FETCH_SIZE is taken uncorrected."""
    bench = _bench()
    v, src = bench.committed_traffic("conv0", "cfg2")
    assert v and "traffic.json" in src and "x2" in src
    v3, src3 = bench.committed_traffic("warp_variance", "cfg3")
    assert v3 and "_cfg3.json" in src3 and "x1" in src3
    assert 1.9e9 < v3 < 2.4e9          # 1.94 GB of bf16 volume writes + the gathers' misses
    assert bench.committed_traffic("softargmin", "cfg2") == (None, None)


def test_bench_live_traffic_parses_pmc_passes_and_degrades_to_none(tmp_path, monkeypatch):
    """bench.py measures roofline.traffic with two rocprofv3 child passes; a stand-in `rocprofv3` on PATH checks the
    command line it is given (one counter per pass, --kernel-trace only, the program itself after `--`) and the
    parsing (per-launch average of the named kernel, FETCH x2 + WRITE, KiB units); a failing one gives None."""
    import stat
    bench = _bench()
    fake = tmp_path / "rocprofv3"
    fake.write_text("""#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
assert a[0] == "--pmc" and a[2] == "--kernel-trace" and "--sys-trace" not in a and "python" in os.path.basename(a[a.index("--") + 1])
ctr, out = a[1], a[a.index("-d") + 1]
if os.environ.get("FAKE_FAIL"):
    sys.exit(3)
os.makedirs(os.path.join(out, "host"), exist_ok=True)
val = {"FETCH_SIZE": [1000.0, 3000.0], "WRITE_SIZE": [500.0, 500.0]}[ctr]
with open(os.path.join(out, "host", "1_counter_collection.csv"), "w") as f:
    f.write("Kernel_Name,Counter_Name,Counter_Value\\n")
    for v in val:
        f.write(f'"void mvs::conv0_w43_mfma_kernel<0>(x)",{ctr},{v}\\n')
    f.write(f'"void mvs::other_kernel(x)",{ctr},777777\\n')
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    assert bench.live_traffic("conv0_w43_mfma_kernel", "conv0", reps=1, timeout=30) == int((2 * 2000.0 + 500.0) * 1024)
    # the warp kernel's scattered tap gathers are NOT under-reported: its FETCH_SIZE counts once
    assert bench.TRAFFIC_KERNELS["warp_variance"][2] == 1.0 and bench.TRAFFIC_KERNELS["conv0"][2] == 2.0
    assert bench.live_traffic("conv0_w43_mfma_kernel", "conv0", reps=1, timeout=30, cfg="cfg3", storage="bf16",
                              fetch_factor=1.0) == int((2000.0 + 500.0) * 1024)
    assert bench.live_traffic("no_such_kernel", "conv0", reps=1, timeout=30) is None
    monkeypatch.setenv("FAKE_FAIL", "1")
    assert bench.live_traffic("conv0_w43_mfma_kernel", "conv0", reps=1, timeout=30) is None
