"""ctypes front-end of the CPU parity oracle (oracle/mvs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by scene_3dreconstruction_mvsnet_amd (the product).

Parity pin: tests/test_oracle_golden.py checks every function here against vectors captured
from the imported reference (tests/golden/).  All arrays are float32, reference layout (NCDHW).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmvs_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (no-op if the .so is newer than the source)."""
    src = os.path.join(_HERE, "mvs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmvs_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_F)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(ctypes.c_int(n))


def relative_proj(src_proj: np.ndarray, ref_proj: np.ndarray) -> np.ndarray:
    """rt[12] = rows 0..2 of src_proj @ inverse(ref_proj): rot (9, row-major) then trans (3).

    models/module.py:107-109.  float32 LAPACK inverse, as torch.inverse on CPU.
    """
    proj = _f32(src_proj) @ np.linalg.inv(_f32(ref_proj)).astype(np.float32)
    proj = proj.astype(np.float32)
    return np.concatenate([proj[:3, :3].reshape(9), proj[:3, 3]]).astype(np.float32)


def homo_warp(src_fea, src_proj, ref_proj, depth_values):
    """models/module.py:96-139 for one batch item: [C,h,w] -> [C,D,h,w]."""
    fea = _f32(src_fea)
    C, h, w = fea.shape
    dv = _f32(depth_values)
    D = dv.shape[0]
    rt = relative_proj(src_proj, ref_proj)
    out = np.empty((C, D, h, w), np.float32)
    lib().orc_homo_warp(_p(fea), _p(rt), _p(dv), _p(out), C, D, h, w)
    return out


def variance_volume(features, proj_matrices, depth_values):
    """models/mvsnet.py:145-177 for one batch item: features [N,C,h,w] -> var [C,D,h,w]."""
    feats = _f32(features)
    N, C, h, w = feats.shape
    dv = _f32(depth_values)
    D = dv.shape[0]
    rts = np.stack([relative_proj(proj_matrices[v], proj_matrices[0]) for v in range(1, N)]) \
        if N > 1 else np.zeros((0, 12), np.float32)
    rts = _f32(rts)
    var = np.empty((C, D, h, w), np.float32)
    s1 = np.empty_like(var)
    s2 = np.empty_like(var)
    lib().orc_variance_volume(_p(feats), _p(rts), _p(dv), _p(var), _p(s1), _p(s2), N, C, D, h, w)
    return var


def conv3d(x, w, bias=None, bn=None, stride=1, relu=True):
    """Conv3d k3 p1 (+bias)(+BN eval)(+ReLU): x [Cin,D,H,W], w [Cout,Cin,3,3,3]."""
    x = _f32(x)
    w = _f32(w)
    Cin, D, H, W = x.shape
    Cout = w.shape[0]
    assert w.shape == (Cout, Cin, 3, 3, 3)
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    y = np.empty((Cout, Do, Ho, Wo), np.float32)
    g, b, m, v = (None,) * 4 if bn is None else [_f32(t) for t in bn]
    bias = None if bias is None else _f32(bias)
    lib().orc_conv3d(_p(x), _p(w), _p(bias), _p(g), _p(b), _p(m), _p(v), _p(y),
                     Cin, Cout, D, H, W, stride, int(relu))
    return y


def conv2d(x, w, bias=None, bn=None, stride=1, relu=True):
    """Conv2d k x k, pad k//2 (+bias)(+BN eval)(+ReLU): x [Cin,H,W], w [Cout,Cin,k,k]
    (models/module.py:6-13)."""
    x = _f32(x)
    w = _f32(w)
    Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    assert w.shape == (Cout, Cin, k, k)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = np.empty((Cout, Ho, Wo), np.float32)
    g, b, m, v = (None,) * 4 if bn is None else [_f32(t) for t in bn]
    bias = None if bias is None else _f32(bias)
    lib().orc_conv2d(_p(x), _p(w), _p(bias), _p(g), _p(b), _p(m), _p(v), _p(y),
                     Cin, Cout, H, W, k, stride, int(relu))
    return y


# (name, stride) of FeatureNet's ConvBnReLU blocks, models/mvsnet.py:15-22
FEATURE_BLOCKS = (("conv0", 1), ("conv1", 1), ("conv2", 2), ("conv3", 1), ("conv4", 1), ("conv5", 2),
                  ("conv6", 1))


def feature_net(img, sd, prefix="feature."):
    """FeatureNet.forward (models/mvsnet.py:26-30) for one image [3,H,W] -> [32,H/4,W/4];
    `sd` = state dict of the whole model (keys `feature.convN.conv.weight`, `feature.convN.bn.*`,
    `feature.feature.{weight,bias}`)."""
    x = _f32(img)
    for name, stride in FEATURE_BLOCKS:
        bn = [sd[f"{prefix}{name}.bn.{k}"] for k in ("weight", "bias", "running_mean", "running_var")]
        x = conv2d(x, sd[f"{prefix}{name}.conv.weight"], bn=bn, stride=stride, relu=True)
    return conv2d(x, sd[f"{prefix}feature.weight"], bias=sd[f"{prefix}feature.bias"], relu=False)


def deconv3d(x, w, bn=None, relu=True):
    """ConvTranspose3d k3 s2 p1 op1 (+BN eval)(+ReLU): x [Cin,D,H,W], w [Cin,Cout,3,3,3]."""
    x = _f32(x)
    w = _f32(w)
    Cin, D, H, W = x.shape
    Cout = w.shape[1]
    assert w.shape == (Cin, Cout, 3, 3, 3)
    y = np.empty((Cout, 2 * D, 2 * H, 2 * W), np.float32)
    g, b, m, v = (None,) * 4 if bn is None else [_f32(t) for t in bn]
    lib().orc_deconv3d(_p(x), _p(w), _p(g), _p(b), _p(m), _p(v), _p(y), Cin, Cout, D, H, W,
                       int(relu))
    return y


def _bn(sd, prefix):
    return (sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"],
            sd[prefix + ".running_var"])


def round_storage(a, storage="f32"):
    """Round an fp32 array to the storage dtype of the HIP path's volumes and back to fp32.

    The reference is fp32 throughout; the 16-bit storage variants (BASELINE.json configs 2/4) keep
    fp32 arithmetic and narrow only what is written between stages.  This gives the
    "quantisation-point-matched" oracle of SURVEY.md §7: same rounding points, same RNE rounding.
    """
    if storage in ("f32", None):
        return a
    a = np.ascontiguousarray(a, dtype=np.float32)
    if storage == "f16":
        return a.astype(np.float16).astype(np.float32)
    if storage == "bf16":
        u = a.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
        return np.where(np.isnan(a), a, out)
    raise ValueError(storage)


def _fold(sd, wkey, bnprefix, transposed=False):
    """BN-folded weights in Conv3d layout [Cout,Cin,3,3,3] and the folded bias (as mvs_pack_weights)."""
    w = np.asarray(sd[wkey], np.float32)
    if transposed:
        w = np.ascontiguousarray(w.transpose(1, 0, 2, 3, 4))
    g, b, m, v = [np.asarray(t, np.float32) for t in _bn(sd, bnprefix)]
    scale = (g / np.sqrt(v + np.float32(1e-5))).astype(np.float32)
    shift = (b - m * scale).astype(np.float32)
    return (w * scale[:, None, None, None, None]).astype(np.float32), shift


def costreg_forward_arith16(var, sd, storage, fused_tail=True):
    """CostRegNet with 16-bit MFMA operands (conv3d_mfma16.hip): BN-folded weights rounded to the
    storage dtype, activations stored in it, fp32 accumulation, fp32 bias/ReLU/skip.
    fused_tail (the HIP path's default since round 3, conv11_prob.hip): conv11's output + skip feeds prob in fp32,
    it is never stored -- the one tensor of the 16-bit modes that is NOT rounded."""
    q = lambda t: round_storage(t, storage)  # noqa: E731

    def conv(x, i, stride=1):
        w, sh = _fold(sd, f"conv{i}.conv.weight", f"conv{i}.bn")
        return q(conv3d(x, q(w), bias=sh, bn=None, stride=stride, relu=True))

    def deconv(x, name, skip, store=True):
        w, sh = _fold(sd, f"{name}.0.weight", f"{name}.1", transposed=True)  # [Cout,Cin,...]
        wt = np.ascontiguousarray(q(w).transpose(1, 0, 2, 3, 4))              # back to [Cin,Cout,...]
        y = deconv3d(x, wt, bn=None, relu=False) + sh[:, None, None, None]
        y = skip + np.maximum(y, 0.0)
        return q(y) if store else y

    c0 = conv(var, 0)
    c2 = conv(conv(c0, 1, 2), 2)
    c4 = conv(conv(c2, 3, 2), 4)
    x = conv(conv(c4, 5, 2), 6)
    x = deconv(x, "conv7", c4)
    x = deconv(x, "conv9", c2)
    x = deconv(x, "conv11", c0, store=not fused_tail)
    cost = conv3d(x, sd["prob.weight"], bias=sd["prob.bias"], bn=None, relu=False)
    return cost[0]


def costreg_forward(var, sd, storage="f32", arith16=False):
    """CostRegNet.forward (models/mvsnet.py:64-73): var [32,D,h,w] -> cost [D,h,w].

    `sd` maps reference parameter names *relative to cost_regularization* to numpy arrays.
    `storage` != "f32" rounds every stored activation (see round_storage); logits stay fp32.
    `arith16` additionally rounds the BN-folded weights (the HIP path's 16-bit MFMA mode).
    """
    if arith16 and storage not in ("f32", None):
        return costreg_forward_arith16(var, sd, storage)
    q = lambda t: round_storage(t, storage)  # noqa: E731
    c0 = q(conv3d(var, sd["conv0.conv.weight"], bn=_bn(sd, "conv0.bn")))
    c1 = q(conv3d(c0, sd["conv1.conv.weight"], bn=_bn(sd, "conv1.bn"), stride=2))
    c2 = q(conv3d(c1, sd["conv2.conv.weight"], bn=_bn(sd, "conv2.bn")))
    c3 = q(conv3d(c2, sd["conv3.conv.weight"], bn=_bn(sd, "conv3.bn"), stride=2))
    c4 = q(conv3d(c3, sd["conv4.conv.weight"], bn=_bn(sd, "conv4.bn")))
    c5 = q(conv3d(c4, sd["conv5.conv.weight"], bn=_bn(sd, "conv5.bn"), stride=2))
    c6 = q(conv3d(c5, sd["conv6.conv.weight"], bn=_bn(sd, "conv6.bn")))
    x = q(c4 + deconv3d(c6, sd["conv7.0.weight"], bn=_bn(sd, "conv7.1")))
    x = q(c2 + deconv3d(x, sd["conv9.0.weight"], bn=_bn(sd, "conv9.1")))
    x = q(c0 + deconv3d(x, sd["conv11.0.weight"], bn=_bn(sd, "conv11.1")))
    cost = conv3d(x, sd["prob.weight"], bias=sd["prob.bias"], bn=None, relu=False)
    return cost[0]


def softargmin_conf(cost, depth_values, want_prob=False):
    """models/mvsnet.py:192-218: cost [D,h,w] -> (depth [h,w], conf [h,w], exp_index[, prob])."""
    cost = _f32(cost)
    D, h, w = cost.shape
    dv = _f32(depth_values)
    depth = np.empty((h, w), np.float32)
    conf = np.empty((h, w), np.float32)
    idx = np.empty((h, w), np.float32)
    prob = np.empty((D, h, w), np.float32) if want_prob else None
    lib().orc_softargmin_conf(_p(cost), _p(dv), _p(depth), _p(conf), _p(prob), _p(idx), D, h, w)
    if want_prob:
        return depth, conf, idx, prob
    return depth, conf, idx


def costreg_state(full_state: dict) -> dict:
    """Strip `module.` / `cost_regularization.` prefixes from a reference state dict."""
    out = {}
    for k, v in full_state.items():
        k = k[len("module."):] if k.startswith("module.") else k
        if k.startswith("cost_regularization."):
            out[k[len("cost_regularization."):]] = np.asarray(v, dtype=np.float32)
    return out


def depth_infer(features, proj_matrices, depth_values, sd, storage="f32", arith16=True, feat16=False):
    """The whole hot path after FeatureNet for one batch item (models/mvsnet.py:145-218).
    With 16-bit storage the defaults match the HIP path's defaults: 16-bit MFMA operands, and (since round 4) the fp32
    features for the warp gather; feat16=True = a 16-bit copy of the features (the HIP path's MVS_FEAT16=1)."""
    feats = round_storage(_f32(features), storage) if feat16 else features
    var = round_storage(variance_volume(feats, proj_matrices, depth_values), storage)
    cost = costreg_forward(var, sd, storage, arith16=arith16)
    depth, conf, idx = softargmin_conf(cost, depth_values)
    return depth, conf
