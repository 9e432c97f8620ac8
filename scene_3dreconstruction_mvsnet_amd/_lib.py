"""ctypes binding of libmvs_hip.so (the C ABI declared in include/mvs_abi.h).

There is no CPU fallback: if the shared library is missing or an entry point fails, a
RuntimeError is raised.  `import torch` must happen before the library is loaded so that the
HIP runtime torch ships (same soname, libamdhip64.so.7) is the one both sides share -- stream
handles and device pointers are only meaningful inside one runtime instance.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch  # noqa: F401  (must be imported first, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVS_LIB_PATH") or os.path.join(_HERE, "csrc", "libmvs_hip.so")

MVS_F32, MVS_F16, MVS_BF16 = 0, 1, 2
DTYPE_CODES = {"f32": MVS_F32, "f16": MVS_F16, "bf16": MVS_BF16,
               "float32": MVS_F32, "float16": MVS_F16, "bfloat16": MVS_BF16}
TORCH_DTYPES = {MVS_F32: torch.float32, MVS_F16: torch.float16, MVS_BF16: torch.bfloat16}


def dtype_code(dtype) -> int:
    """'f32' | 'f16' | 'bf16' | torch dtype | mvs_dtype int -> mvs_dtype int (storage dtype of the
    private volumes; arithmetic stays fp32)."""
    if isinstance(dtype, int):
        if dtype in TORCH_DTYPES:
            return dtype
        raise ValueError(f"unknown mvs_dtype {dtype}")
    if isinstance(dtype, torch.dtype):
        for code, td in TORCH_DTYPES.items():
            if td == dtype:
                return code
        raise ValueError(f"unsupported storage dtype {dtype}")
    return DTYPE_CODES[str(dtype)]
NUM_LAYERS = 11
ABI_VERSION = 2

# every symbol include/mvs_abi.h declares
SYMBOLS = (
    "mvs_abi_version", "mvs_last_error_string", "mvs_query_workspace", "mvs_query_weights_blob",
    "mvs_pack_weights", "mvs_relative_proj", "mvs_warp_variance", "mvs_costreg_forward",
    "mvs_conv_layer", "mvs_conv11_prob", "mvs_softargmin_conf", "mvs_depth_infer", "mvs_homo_warp", "mvs_depth_regression",
    "mvs_filter_compose", "mvs_filter_depth",
    "mvs_query_feature_blob", "mvs_pack_feature_weights", "mvs_query_feature_workspace",
    "mvs_feature_layer", "mvs_feature_net", "mvs_query_forward_workspace", "mvs_forward_images",
    "mvs_feature_net_fmt", "mvs_forward_images_fmt",
)

# mvs_image_format (include/mvs_abi.h)
MVS_IMG_F32_CHW, MVS_IMG_U8_CHW, MVS_IMG_U8_HWC = 0, 1, 2

_lock = threading.Lock()
_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_sz = ctypes.c_size_t


class MvsError(RuntimeError):
    """Non-zero status from libmvs_hip.so (decoded with mvs_last_error_string)."""

    def __init__(self, code, msg):
        super().__init__(f"libmvs_hip status {code}: {msg}")
        self.code = code


def load():
    """Load libmvs_hip.so once; raises RuntimeError if it is missing (no fallback)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or make -C scene_3dreconstruction_mvsnet_amd/csrc). "
                "There is no CPU/PyTorch fallback for the MVSNet depth path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name in SYMBOLS:
            if not hasattr(lib, name):
                raise RuntimeError(f"{LIB_PATH} does not export {name}")
        lib.mvs_abi_version.restype = _i
        lib.mvs_last_error_string.restype = ctypes.c_char_p
        lib.mvs_query_workspace.argtypes = [_i, _i, _i, _i, _i, _i, ctypes.POINTER(_sz)]
        lib.mvs_query_weights_blob.argtypes = [ctypes.POINTER(_sz)]
        lib.mvs_pack_weights.argtypes = [ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp,
                                         ctypes.c_float, _vp, _sz]
        lib.mvs_relative_proj.argtypes = [_vp, _vp, _i, _vp]
        lib.mvs_warp_variance.argtypes = [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp]
        lib.mvs_costreg_forward.argtypes = [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp]
        lib.mvs_conv_layer.argtypes = [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]
        lib.mvs_conv11_prob.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]
        lib.mvs_softargmin_conf.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]
        lib.mvs_depth_infer.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                        _i, _i, _i, _i, _i, _i, _vp]
        lib.mvs_homo_warp.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]
        lib.mvs_depth_regression.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp]
        lib.mvs_query_feature_blob.argtypes = [ctypes.POINTER(_sz)]
        lib.mvs_pack_feature_weights.argtypes = [ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp,
                                                 ctypes.c_float, _vp, _sz]
        lib.mvs_query_feature_workspace.argtypes = [_i, _i, _i, ctypes.POINTER(_sz)]
        lib.mvs_feature_layer.argtypes = [_i, _vp, _vp, _vp, _i, _i, _i, _vp]
        lib.mvs_feature_net.argtypes = [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]
        lib.mvs_query_forward_workspace.argtypes = [_i, _i, _i, _i, _i, ctypes.POINTER(_sz)]
        lib.mvs_forward_images.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                           _i, _i, _i, _i, _i, _vp]
        lib.mvs_feature_net_fmt.argtypes = [_vp, _i, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]
        lib.mvs_forward_images_fmt.argtypes = [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                               _i, _i, _i, _i, _i, _vp]
        lib.mvs_filter_compose.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]
        _d = ctypes.c_double
        lib.mvs_filter_depth.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                         _d, _i, _d, _d, _vp, _vp, _vp, _vp, _vp]
        for name in SYMBOLS:
            if name not in ("mvs_last_error_string",):
                getattr(lib, name).restype = _i
        if lib.mvs_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libmvs_hip ABI {lib.mvs_abi_version()} != expected {ABI_VERSION}")
        _lib = lib
        return _lib


def check(status: int) -> None:
    if status != 0:
        raise MvsError(status, load().mvs_last_error_string().decode("utf-8", "replace"))


def query_workspace(N, C, D, h, w, dtype=MVS_F32) -> int:
    n = _sz(0)
    check(load().mvs_query_workspace(N, C, D, h, w, dtype, ctypes.byref(n)))
    return int(n.value)


def query_weights_blob() -> int:
    n = _sz(0)
    check(load().mvs_query_weights_blob(ctypes.byref(n)))
    return int(n.value)


def _stream(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); the MVSNet depth path "
                           "has no CPU implementation")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return t.contiguous()


# reference parameter names, relative to `cost_regularization.` (models/mvsnet.py:35-62)
CONV_WEIGHT_KEYS = tuple([f"conv{i}.conv.weight" for i in range(7)] +
                         ["conv7.0.weight", "conv9.0.weight", "conv11.0.weight", "prob.weight"])
BN_PREFIXES = tuple([f"conv{i}.bn" for i in range(7)] + ["conv7.1", "conv9.1", "conv11.1"])
_LAYER_CH = ((32, 8), (8, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64),
             (64, 32), (32, 16), (16, 8), (8, 1))


def pack_weights(state: dict, eps: float = 1e-5) -> torch.Tensor:
    """BN-fold + re-layout the CostRegNet parameters into the kernels' blob (host, uint8).

    `state` maps names relative to `cost_regularization.` to CPU float32 tensors/arrays.
    """
    import numpy as np

    lib = load()
    keep = []

    def arr(key, shape):
        a = state[key]
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        a = np.ascontiguousarray(a, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise RuntimeError(f"{key}: shape {tuple(a.shape)} != expected {tuple(shape)}")
        keep.append(a)
        return a.ctypes.data

    convs = (_vp * NUM_LAYERS)()
    for l, key in enumerate(CONV_WEIGHT_KEYS):
        ci, co = _LAYER_CH[l]
        shape = (ci, co, 3, 3, 3) if 7 <= l <= 9 else (co, ci, 3, 3, 3)
        convs[l] = arr(key, shape)
    bns = (_vp * 40)()
    for l, pre in enumerate(BN_PREFIXES):
        co = _LAYER_CH[l][1]
        for j, suffix in enumerate(("weight", "bias", "running_mean", "running_var")):
            bns[4 * l + j] = arr(f"{pre}.{suffix}", (co,))
    bias = arr("prob.bias", (1,))
    nbytes = query_weights_blob()
    blob = torch.empty(nbytes, dtype=torch.uint8)
    check(lib.mvs_pack_weights(convs, bns, bias, ctypes.c_float(eps), blob.data_ptr(), nbytes))
    return blob


def relative_proj(proj: torch.Tensor) -> torch.Tensor:
    """proj [N,4,4] cuda -> rt [(N-1),12] cuda (models/module.py:107-109)."""
    proj = _dev_f32(proj, "proj_matrices")
    N = proj.shape[0]
    rt = torch.empty((max(N - 1, 1), 12), dtype=torch.float32, device=proj.device)
    check(load().mvs_relative_proj(proj.data_ptr(), rt.data_ptr(), N, _stream(proj.device)))
    return rt


def warp_variance(feats, rt, depth_values, workspace, dtype=MVS_F32):
    """feats [N,32,h,w], rt [(N-1),12], depth_values [D] -> C8-planar volume [4,D,h,w,8]."""
    feats = _dev_f32(feats, "features")
    N, C, h, w = feats.shape
    D = depth_values.shape[0]
    var = torch.empty((C // 8, D, h, w, 8), dtype=TORCH_DTYPES[dtype], device=feats.device)
    check(load().mvs_warp_variance(feats.data_ptr(), rt.data_ptr(),
                                   _dev_f32(depth_values, "depth_values").data_ptr(),
                                   var.data_ptr(), workspace.data_ptr(), workspace.numel(),
                                   N, C, D, h, w, dtype, _stream(feats.device)))
    return var


def costreg_forward(var, blob, workspace, dtype=MVS_F32):
    """var [4,D,h,w,8] (C8-planar) -> cost logits [D,h,w] fp32."""
    if var.dtype != TORCH_DTYPES[dtype] or not var.is_cuda or not var.is_contiguous():
        raise RuntimeError(f"variance volume must be a contiguous CUDA tensor of {TORCH_DTYPES[dtype]}")
    _, D, h, w, _ = var.shape
    cost = torch.empty((D, h, w), dtype=torch.float32, device=var.device)
    check(load().mvs_costreg_forward(var.data_ptr(), blob.data_ptr(), cost.data_ptr(),
                                     workspace.data_ptr(), workspace.numel(), D, h, w, dtype,
                                     _stream(var.device)))
    return cost


def conv_layer(layer, x, skip, blob, dtype=MVS_F32):
    """One CostRegNet layer on C8-planar tensors [Cin/8,D,h,w,8] -> [Cout/8,D',h',w',8]."""
    ci, co = _LAYER_CH[layer]
    planes, Di, Hi, Wi, c8 = x.shape
    if planes * c8 != ci or c8 != 8:
        raise RuntimeError(f"layer {layer}: input has {planes}x{c8} channels, expected {ci}")
    if x.dtype != TORCH_DTYPES[dtype] or (skip is not None and skip.dtype != TORCH_DTYPES[dtype]):
        raise RuntimeError(f"layer {layer}: tensors are {x.dtype}, storage dtype says {TORCH_DTYPES[dtype]}")
    if not x.is_cuda or not x.is_contiguous() or (skip is not None and not skip.is_contiguous()):
        raise RuntimeError(f"layer {layer}: needs contiguous CUDA(ROCm) tensors")
    if 7 <= layer <= 9:
        odims = (2 * Di, 2 * Hi, 2 * Wi)
    elif layer in (1, 3, 5):
        odims = ((Di - 1) // 2 + 1, (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1)
    else:
        odims = (Di, Hi, Wi)
    oshape = odims if layer == 10 else (co // 8,) + odims + (8,)
    y = torch.empty(oshape, dtype=torch.float32 if layer == 10 else TORCH_DTYPES[dtype], device=x.device)
    if skip is not None and tuple(skip.shape) != tuple(oshape):
        raise RuntimeError(f"layer {layer}: skip shape {tuple(skip.shape)} != {oshape}")
    check(load().mvs_conv_layer(layer, x.data_ptr(), 0 if skip is None else skip.data_ptr(),
                                y.data_ptr(), blob.data_ptr(), Di, Hi, Wi, dtype,
                                _stream(x.device)))
    return y


def conv11_prob(x, skip, blob, dtype=MVS_F32):
    """Layers 9 + 10 in one kernel: x [2,Di,Hi,Wi,8], skip [1,2Di,2Hi,2Wi,8] (storage dtype) -> fp32 cost logits
    [2Di,2Hi,2Wi]."""
    planes, Di, Hi, Wi, c8 = x.shape
    if planes != 2 or c8 != 8 or tuple(skip.shape) != (1, 2 * Di, 2 * Hi, 2 * Wi, 8):
        raise RuntimeError(f"conv11_prob: shapes {tuple(x.shape)} / {tuple(skip.shape)}")
    if x.dtype != TORCH_DTYPES[dtype] or skip.dtype != TORCH_DTYPES[dtype]:
        raise RuntimeError(f"conv11_prob: tensors must be {TORCH_DTYPES[dtype]}")
    if not (x.is_cuda and skip.is_cuda and x.is_contiguous() and skip.is_contiguous()):
        raise RuntimeError("conv11_prob: needs contiguous CUDA(ROCm) tensors")
    cost = torch.empty((2 * Di, 2 * Hi, 2 * Wi), dtype=torch.float32, device=x.device)
    check(load().mvs_conv11_prob(x.data_ptr(), skip.data_ptr(), cost.data_ptr(), blob.data_ptr(), Di, Hi, Wi,
                                 dtype, _stream(x.device)))
    return cost


def softargmin_conf(cost, depth_values):
    cost = _dev_f32(cost, "cost")
    D, h, w = cost.shape
    depth = torch.empty((h, w), dtype=torch.float32, device=cost.device)
    conf = torch.empty_like(depth)
    check(load().mvs_softargmin_conf(cost.data_ptr(),
                                     _dev_f32(depth_values, "depth_values").data_ptr(),
                                     depth.data_ptr(), conf.data_ptr(), D, h, w,
                                     _stream(cost.device)))
    return depth, conf


def depth_infer(feats, proj, depth_values, blob, workspace, depth_out, conf_out, dtype=MVS_F32):
    """Whole path for one batch item; outputs are written into depth_out / conf_out [h,w]."""
    N, C, h, w = feats.shape
    D = depth_values.shape[0]
    check(load().mvs_depth_infer(feats.data_ptr(), proj.data_ptr(), depth_values.data_ptr(),
                                 blob.data_ptr(), depth_out.data_ptr(), conf_out.data_ptr(),
                                 workspace.data_ptr(), workspace.numel(), N, C, D, h, w, dtype,
                                 _stream(feats.device)))


def alloc_workspace(N, C, D, h, w, device, dtype=MVS_F32) -> torch.Tensor:
    nbytes = query_workspace(N, C, D, h, w, dtype)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def to_c8(t: torch.Tensor) -> torch.Tensor:
    """[C,D,h,w] (reference layout, one batch item) -> C8-planar [C/8,D,h,w,8]."""
    C = t.shape[0]
    return t.reshape(C // 8, 8, *t.shape[1:]).permute(0, 2, 3, 4, 1).contiguous()


def from_c8(t: torch.Tensor) -> torch.Tensor:
    """C8-planar [C/8,D,h,w,8] -> [C,D,h,w]."""
    P, D, h, w, _ = t.shape
    return t.permute(0, 4, 1, 2, 3).reshape(P * 8, D, h, w).contiguous()


FILTER_REF_FLOATS, FILTER_PAIR_FLOATS = 30, 42


def filter_compose(intrinsics, extrinsics, ref_idx, src_idx):
    """Host-side float32 camera products for mvs_filter_depth (numpy arrays in / out).

    intrinsics [V,3,3], extrinsics [V,4,4] float32; ref_idx [R] int32; src_idx [R,S] int32 (-1 pad).
    """
    import numpy as np
    K = np.ascontiguousarray(intrinsics, np.float32).reshape(-1, 9)
    E = np.ascontiguousarray(extrinsics, np.float32).reshape(-1, 16)
    ref = np.ascontiguousarray(ref_idx, np.int32)
    src = np.ascontiguousarray(src_idx, np.int32)
    if K.shape[0] != E.shape[0] or src.ndim != 2 or src.shape[0] != ref.shape[0]:
        raise RuntimeError(f"filter_compose: inconsistent shapes K{K.shape} E{E.shape} ref{ref.shape} src{src.shape}")
    V, R, S = K.shape[0], ref.shape[0], src.shape[1]
    ref_mats = np.empty((R, FILTER_REF_FLOATS), np.float32)
    pair_mats = np.empty((R, S, FILTER_PAIR_FLOATS), np.float32)
    check(load().mvs_filter_compose(K.ctypes.data, E.ctypes.data, ref.ctypes.data, src.ctypes.data,
                                    V, R, S, ref_mats.ctypes.data, pair_mats.ctypes.data))
    return ref_mats, pair_mats


def filter_depth(depth, conf, ref_mats, pair_mats, ref_idx, src_idx, photomask=0.8, geomask=3,
                 condmask_pixel=1.0, condmask_depth=0.01):
    """Device tensors in, device tensors out: geo_sum int32 [R,h,w], depth_avg float64 [R,h,w],
    masks uint8 [R,3,h,w] (photo, geo, final), xyz_world float64 [R,h*w,3]."""
    depth = _dev_f32(depth, "depth")
    conf = _dev_f32(conf, "conf")
    dev = depth.device
    if depth.dim() != 3 or conf.shape != depth.shape:
        raise RuntimeError(f"filter_depth: depth {tuple(depth.shape)} / conf {tuple(conf.shape)} must both be [V,h,w]")
    V, h, w = depth.shape
    ref_mats = _dev_f32(ref_mats, "ref_mats")
    pair_mats = _dev_f32(pair_mats, "pair_mats")
    if ref_idx.dtype != torch.int32 or src_idx.dtype != torch.int32 or not ref_idx.is_cuda or not src_idx.is_cuda:
        raise RuntimeError("filter_depth: ref_idx / src_idx must be int32 tensors on the GPU")
    R, S = src_idx.shape
    if tuple(ref_mats.shape) != (R, FILTER_REF_FLOATS) or tuple(pair_mats.shape) != (R, S, FILTER_PAIR_FLOATS) \
            or ref_idx.numel() != R:
        raise RuntimeError("filter_depth: ref_mats / pair_mats / ref_idx do not match src_idx's [R,S]")
    ref_idx, src_idx = ref_idx.contiguous(), src_idx.contiguous()
    geo = torch.empty((R, h, w), dtype=torch.int32, device=dev)
    avg = torch.empty((R, h, w), dtype=torch.float64, device=dev)
    masks = torch.empty((R, 3, h, w), dtype=torch.uint8, device=dev)
    xyz = torch.empty((R, h * w, 3), dtype=torch.float64, device=dev)
    check(load().mvs_filter_depth(depth.data_ptr(), conf.data_ptr(), ref_mats.data_ptr(),
                                  pair_mats.data_ptr(), ref_idx.data_ptr(), src_idx.data_ptr(),
                                  V, R, S, h, w, float(photomask), int(geomask), float(condmask_pixel),
                                  float(condmask_depth), geo.data_ptr(), avg.data_ptr(), masks.data_ptr(),
                                  xyz.data_ptr(), _stream(dev)))
    return geo, avg, masks, xyz


# ---- FeatureNet (reference models/mvsnet.py:10-30) -------------------------------------------
# (cin, cout, k, stride) of conv0..conv6 and the final `feature` Conv2d
FEATURE_LAYERS = ((3, 8, 3, 1), (8, 8, 3, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 16, 3, 1),
                  (16, 32, 5, 2), (32, 32, 3, 1), (32, 32, 3, 1))
FEATURE_WEIGHT_KEYS = tuple([f"conv{i}.conv.weight" for i in range(7)] + ["feature.weight"])


def query_feature_blob() -> int:
    n = _sz(0)
    check(load().mvs_query_feature_blob(ctypes.byref(n)))
    return int(n.value)


def pack_feature_weights(state: dict, eps: float = 1e-5) -> torch.Tensor:
    """BN-fold + re-layout FeatureNet's parameters into MFMA panels (host, uint8).
    `state` maps names relative to `feature.` to CPU float32 tensors/arrays."""
    import numpy as np
    keep = []

    def arr(key, shape):
        a = state[key]
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        a = np.ascontiguousarray(a, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise RuntimeError(f"{key}: shape {tuple(a.shape)} != expected {tuple(shape)}")
        keep.append(a)
        return a.ctypes.data

    convs = (_vp * 8)()
    for l, key in enumerate(FEATURE_WEIGHT_KEYS):
        ci, co, k, _ = FEATURE_LAYERS[l]
        convs[l] = arr(key, (co, ci, k, k))
    bns = (_vp * 28)()
    for l in range(7):
        co = FEATURE_LAYERS[l][1]
        for j, suffix in enumerate(("weight", "bias", "running_mean", "running_var")):
            bns[4 * l + j] = arr(f"conv{l}.bn.{suffix}", (co,))
    bias = arr("feature.bias", (32,))
    nbytes = query_feature_blob()
    blob = torch.empty(nbytes, dtype=torch.uint8)
    check(load().mvs_pack_feature_weights(convs, bns, bias, ctypes.c_float(eps), blob.data_ptr(), nbytes))
    return blob


def query_feature_workspace(N, H, W) -> int:
    n = _sz(0)
    check(load().mvs_query_feature_workspace(N, H, W, ctypes.byref(n)))
    return int(n.value)


def query_forward_workspace(N, H, W, D, dtype=MVS_F32) -> int:
    n = _sz(0)
    check(load().mvs_query_forward_workspace(N, H, W, D, dtype, ctypes.byref(n)))
    return int(n.value)


def feature_layer(layer, x, fblob):
    """One FeatureNet layer on the GPU.  x: NCHW images [N,3,H,W] for layer 0, otherwise C8-planar
    [Cin/8,N,H,W,8]; returns C8-planar [Cout/8,N,Ho,Wo,8]."""
    x = _dev_f32(x, "x")
    ci, co, k, s = FEATURE_LAYERS[layer]
    if layer == 0:
        N, c, H, W = x.shape
        if c != 3:
            raise RuntimeError(f"feature layer 0 wants [N,3,H,W] images, got {tuple(x.shape)}")
    else:
        pl, N, H, W, e = x.shape
        if pl * e != ci or e != 8:
            raise RuntimeError(f"feature layer {layer} wants C8-planar input with {ci} channels, got {tuple(x.shape)}")
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    y = torch.empty((co // 8, N, Ho, Wo, 8), dtype=torch.float32, device=x.device)
    check(load().mvs_feature_layer(layer, x.data_ptr(), y.data_ptr(), fblob.data_ptr(), N, H, W,
                                   _stream(x.device)))
    return y


def _image_arg(imgs, what):
    """Device images in one of the ABI's pixel formats -> (tensor, mvs_image_format, N, H, W):
    float32 [N,3,H,W]; uint8 [N,3,H,W]; uint8 [N,H,W,3] (as PIL yields a decoded image).  The uint8 forms are
    divided by 255 inside FeatureNet's first kernel -- the reference's host-side conversion
    (datasets/data_io.py:143), bit for bit -- so the caller copies a quarter of the bytes."""
    if not isinstance(imgs, torch.Tensor) or not imgs.is_cuda:
        raise RuntimeError(f"{what}: images must be a CUDA(ROCm) tensor")
    if imgs.dim() != 4:
        raise RuntimeError(f"{what} wants [N,3,H,W] (float32 / uint8) or [N,H,W,3] (uint8) images, got {tuple(imgs.shape)}")
    if imgs.dtype == torch.uint8:
        imgs = imgs.contiguous()
        if imgs.shape[1] == 3:
            return imgs, MVS_IMG_U8_CHW, imgs.shape[0], imgs.shape[2], imgs.shape[3]
        if imgs.shape[3] == 3:
            return imgs, MVS_IMG_U8_HWC, imgs.shape[0], imgs.shape[1], imgs.shape[2]
        raise RuntimeError(f"{what}: uint8 images must be [N,3,H,W] or [N,H,W,3], got {tuple(imgs.shape)}")
    imgs = _dev_f32(imgs, "imgs")
    if imgs.shape[1] != 3:
        raise RuntimeError(f"{what} wants [N,3,H,W] images, got {tuple(imgs.shape)}")
    return imgs, MVS_IMG_F32_CHW, imgs.shape[0], imgs.shape[2], imgs.shape[3]


def feature_net(imgs, fblob, workspace=None):
    """FeatureNet.forward on the GPU: imgs [N,3,H,W] fp32 (or uint8, see _image_arg) -> [N,32,H/4,W/4] fp32 (NCHW)."""
    imgs, fmt, N, H, W = _image_arg(imgs, "feature_net")
    nbytes = query_feature_workspace(N, H, W)
    if workspace is None:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=imgs.device)
    h4, w4 = ((H - 1) // 2 + 1 - 1) // 2 + 1, ((W - 1) // 2 + 1 - 1) // 2 + 1
    out = torch.empty((N, 32, h4, w4), dtype=torch.float32, device=imgs.device)
    check(load().mvs_feature_net_fmt(imgs.data_ptr(), fmt, fblob.data_ptr(), out.data_ptr(), workspace.data_ptr(),
                                     workspace.numel(), N, H, W, _stream(imgs.device)))
    return out


def forward_images(imgs, proj, depth_values, fblob, blob, workspace, depth_out, conf_out, dtype=MVS_F32):
    """MVSNet.forward of one batch item from images: imgs [N,3,H,W] fp32 (or uint8, see _image_arg), proj [N,4,4],
    depth_values [D]."""
    imgs, fmt, N, H, W = _image_arg(imgs, "forward_images")
    proj = _dev_f32(proj, "proj_matrices")
    depth_values = _dev_f32(depth_values, "depth_values")
    if tuple(proj.shape) != (N, 4, 4):
        raise RuntimeError(f"forward_images: imgs {tuple(imgs.shape)} / proj {tuple(proj.shape)}")
    D = depth_values.numel()
    if tuple(depth_out.shape) != (H // 4, W // 4) or tuple(conf_out.shape) != (H // 4, W // 4) \
            or not depth_out.is_contiguous() or not conf_out.is_contiguous():
        raise RuntimeError("forward_images: depth_out / conf_out must be contiguous [H/4, W/4] float32")
    check(load().mvs_forward_images_fmt(imgs.data_ptr(), fmt, proj.data_ptr(), depth_values.data_ptr(),
                                        fblob.data_ptr(), blob.data_ptr(), depth_out.data_ptr(),
                                        conf_out.data_ptr(), workspace.data_ptr(), workspace.numel(),
                                        N, H, W, D, dtype, _stream(imgs.device)))
