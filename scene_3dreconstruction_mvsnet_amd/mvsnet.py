"""Drop-in `MVSNet` for the reference's models/mvsnet.py (inference path, MI355X-native).

`MVSNet(refine, debug).forward(imgs, proj_matrices, depth_values)` keeps the reference signature
and return dict (reference models/mvsnet.py:91-239); parameter/buffer names equal the
reference's so `load_state_dict(torch.load(ckpt)['model'])` works, with or without the
`module.` prefix that `nn.DataParallel` adds (eval.py:309-315).

What runs where: everything -- FeatureNet (mvsnet.py:10-30), homography warp, variance volume,
CostRegNet, softmax / soft-argmin / photometric confidence -- runs in hand-written gfx950 HIP
kernels behind the C ABI (include/mvs_abi.h -> csrc/libmvs_hip.so), enqueued on torch's current
stream, one mvs_forward_images per batch item.  `model.feature_impl = "torch"` keeps FeatureNet on
PyTorch-ROCm (MIOpen) and hands NCHW features to mvs_depth_infer (the round-1 arrangement; kept
for A/B timing).  The `feature` sub-module is a parameter container either way.

Inference only: forward() raises in training mode, on CPU tensors, or if libmvs_hip.so is
missing -- there is deliberately no PyTorch fallback for the HIP path.
"""
import threading
import warnings

import torch
import torch.nn as nn

from . import _lib
from .module import ConvBnReLU, ConvBnReLU3D


# (in, out, kernel, stride, pad) of the 2D feature extractor's ConvBnReLU blocks conv0..conv6
_FEATURE_BLOCKS = ((3, 8, 3, 1, 1), (8, 8, 3, 1, 1), (8, 16, 5, 2, 2), (16, 16, 3, 1, 1),
                   (16, 16, 3, 1, 1), (16, 32, 5, 2, 2), (32, 32, 3, 1, 1))
# (in, out, stride) of CostRegNet's ConvBnReLU3D blocks conv0..conv6 and (in, out) of its
# ConvTranspose3d+BN+ReLU blocks conv7/conv9/conv11 -- the same table as csrc/mvs_internal.h
_COSTREG_CONVS = ((32, 8, 1), (8, 16, 2), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 64, 2), (64, 64, 1))
_COSTREG_DECONVS = (("conv7", 64, 32), ("conv9", 32, 16), ("conv11", 16, 8))


class FeatureNet(nn.Module):
    """2D feature extractor, unchanged torch ops (reference models/mvsnet.py:10-30): seven
    ConvBnReLU blocks (two of them 5x5 stride 2 -> 1/4 resolution) and a biased 3x3 conv to 32
    channels; sub-module names conv0..conv6 / feature fix the checkpoint keys."""

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        for i, spec in enumerate(_FEATURE_BLOCKS):
            setattr(self, f"conv{i}", ConvBnReLU(*spec))
        self.feature = nn.Conv2d(32, 32, 3, 1, 1)

    def forward(self, x):
        for i in range(len(_FEATURE_BLOCKS)):
            x = getattr(self, f"conv{i}")(x)
        return self.feature(x)


class CostRegNet(nn.Module):
    """Parameter container with the reference's names/shapes (models/mvsnet.py:33-62)."""

    def __init__(self):
        super().__init__()
        for i, (cin, cout, stride) in enumerate(_COSTREG_CONVS):
            setattr(self, f"conv{i}", ConvBnReLU3D(cin, cout, stride=stride))
        for name, cin, cout in _COSTREG_DECONVS:
            setattr(self, name, nn.Sequential(
                nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2,
                                   bias=False),
                nn.BatchNorm3d(cout), nn.ReLU(inplace=True)))
        self.prob = nn.Conv3d(8, 1, 3, stride=1, padding=1)

    def forward(self, x):  # pragma: no cover - never called on the product path
        raise RuntimeError("CostRegNet runs inside libmvs_hip (mvs_costreg_forward)")

    def bn_eps(self):
        return float(self.conv0.bn.eps)


class RefineNet(nn.Module):
    """Parameter container only; the reference's RefineNet.forward is broken (mvsnet.py:85,238)."""

    def __init__(self):
        super().__init__()
        self.conv1 = ConvBnReLU(4, 32)
        self.conv2 = ConvBnReLU(32, 32)
        self.conv3 = ConvBnReLU(32, 32)
        self.res = ConvBnReLU(32, 1)


class MVSNet(nn.Module):
    ACCEPTS_UINT8_IMAGES = True   # forward() also takes uint8 [B,N,3,H,W] / [B,N,H,W,3] images (see forward)

    def __init__(self, refine=True, debug=0):
        super().__init__()
        self.refine = refine
        self.debug = debug
        print('[MVSNet] init (debug={})'.format(self.debug))
        if debug:
            warnings.warn("debug visualisation bits (cv2.imshow in the reference, mvsnet.py:111-232) "
                          "are ignored by the MI355X path")
        # storage dtype of the HIP path's private volumes: "f32" (default, = the reference's fp32
        # pipeline), "f16" or "bf16" (BASELINE.json configs 4 / 2); arithmetic is always fp32
        self.storage_dtype = "f32"
        # "hip": FeatureNet in libmvs_hip (default); "torch": PyTorch-ROCm convolutions
        self.feature_impl = "hip"
        self.feature = FeatureNet()
        self.cost_regularization = CostRegNet()
        if self.refine:
            self.refine_network = RefineNet()
        # per-device caches; plain attributes so nn.DataParallel replicas share them
        self._cache_lock = threading.Lock()
        self._blob_cache = {}       # device index -> (param versions, device blob tensor)
        self._workspace_cache = {}  # (device index, stream, N, D, h, w, dtype) -> uint8 device tensor

    # The lock and the device caches are process-local: drop them when the module is pickled or
    # deep-copied (torch.save(model), copy.deepcopy) and start the copy with empty caches.
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_cache_lock", "_blob_cache", "_workspace_cache", "_dp_source"):
            state.pop(k, None)
        return state

    # nn.DataParallel runs forward on replicas made by torch.nn.parallel.replicate() (eval.py:309
    # wraps the model even on one GPU).  A replica's parameters are plain attributes -- broadcast
    # copies, not `_parameters` entries -- so `state_dict()` / `parameters()` on it see only the BN
    # buffers, and its tensors change address every forward.  Weights are therefore always packed
    # from the SOURCE module, which every replica remembers here (a `__dict__` entry, not a
    # registered sub-module); the caches are shared with it through the shallow `__dict__` copy.
    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        replica.__dict__["_dp_source"] = self._source()
        return replica

    def _source(self):
        return self.__dict__.get("_dp_source", self)

    def __setstate__(self, state):
        super().__setstate__(state)
        self._cache_lock = threading.Lock()
        self._blob_cache = {}
        self._workspace_cache = {}
        self.__dict__.pop("_dp_source", None)

    # -- caches ------------------------------------------------------------------------------
    @staticmethod
    def _param_versions(mod):
        return tuple((t.data_ptr(), t._version) for t in list(mod.parameters()) + list(mod.buffers()))

    def _packable_state(self, which):
        """CPU copy of the source module's `feature` / `cost_regularization` state dict (never a
        replica's: see _replicate_for_data_parallel)."""
        mod = getattr(self._source(), which)
        return {k: v.detach().cpu() for k, v in mod.state_dict().items()}

    def _feature_blob(self, device):
        key = ("feature", device.index if device.index is not None else torch.cuda.current_device())
        versions = self._param_versions(self._source().feature)
        with self._cache_lock:
            hit = self._blob_cache.get(key)
            if hit is not None and hit[0] == versions:
                return hit[1]
            state = self._packable_state("feature")
            blob = _lib.pack_feature_weights(state, eps=self._source().feature.conv0.bn.eps).to(device)
            self._blob_cache[key] = (versions, blob)
            return blob

    # One workspace per (device, stream, shape): forwards enqueued on different streams (user side
    # streams, DataParallel worker threads) must not share the variance / activation volumes --
    # nothing would order them -- while consecutive forwards on one stream re-use one allocation.
    _MAX_CACHED_WORKSPACES = 8   # least recently used entries beyond this are dropped

    def _cached_workspace(self, key, nbytes_fn, device):
        with self._cache_lock:
            ws = self._workspace_cache.pop(key, None)
            if ws is None:
                ws = torch.empty(nbytes_fn(), dtype=torch.uint8, device=device)
            self._workspace_cache[key] = ws   # (re-)inserted last = most recently used
            while len(self._workspace_cache) > self._MAX_CACHED_WORKSPACES:
                self._workspace_cache.pop(next(iter(self._workspace_cache)))
            return ws

    def _forward_workspace(self, device, N, H, W, D, dtype):
        key = ("fwd", device.index, _lib._stream(device), N, H, W, D, dtype)
        return self._cached_workspace(key, lambda: _lib.query_forward_workspace(N, H, W, D, dtype), device)

    def _weights_blob(self, device):
        key = device.index if device.index is not None else torch.cuda.current_device()
        versions = self._param_versions(self._source().cost_regularization)
        with self._cache_lock:
            hit = self._blob_cache.get(key)
            if hit is not None and hit[0] == versions:
                return hit[1]
            state = self._packable_state("cost_regularization")
            blob = _lib.pack_weights(state, eps=self._source().cost_regularization.bn_eps()).to(device)
            self._blob_cache[key] = (versions, blob)
            return blob

    def _workspace(self, device, N, D, h, w, dtype):
        key = (device.index, _lib._stream(device), N, D, h, w, dtype)
        return self._cached_workspace(key, lambda: _lib.query_workspace(N, 32, D, h, w, dtype), device)

    # -- forward -----------------------------------------------------------------------------
    def forward(self, imgs, proj_matrices, depth_values):
        n_imgs, n_proj = imgs.shape[1], proj_matrices.shape[1]
        assert n_imgs == n_proj, "Different number of images and projection matrices"
        if self.training:
            raise RuntimeError("this MVSNet is the MI355X inference path; call .eval() "
                               "(training, models/mvsnet.py:167-169, is out of scope)")
        if not imgs.is_cuda:
            raise RuntimeError("MVSNet.forward needs CUDA(ROCm) tensors: the depth path has no CPU "
                               "implementation (got imgs on {})".format(imgs.device))
        if self.refine:
            raise NotImplementedError("refine=True: the reference's RefineNet path is broken "
                                      "(F.cat at models/mvsnet.py:85); every working caller passes "
                                      "refine=False (eval.py:308)")
        device = imgs.device
        # uint8 images -- [B,N,3,H,W], or [B,N,H,W,3] as a decoder yields them -- are the reference loader's pixels
        # before `np.array(img, float32) / 255.` (datasets/data_io.py:143); the HIP FeatureNet divides on the device
        u8_hwc = imgs.dtype == torch.uint8 and imgs.shape[2] != 3 and imgs.shape[-1] == 3
        if u8_hwc:
            B, N, H, W, _ = imgs.shape
        else:
            B, N, _, H, W = imgs.shape
        D = depth_values.shape[1]
        if self.feature_impl not in ("hip", "torch"):
            raise RuntimeError(f"feature_impl must be 'hip' or 'torch', got {self.feature_impl!r}")
        if self.feature_impl == "hip":
            if (not u8_hwc and imgs.shape[2] != 3) or H % 32 or W % 32:
                raise RuntimeError(f"imgs must be [B,N,3,H,W] with H, W multiples of 32, got {tuple(imgs.shape)}")
            with torch.cuda.device(device), torch.no_grad():
                imgs_f = imgs.contiguous() if imgs.dtype == torch.uint8 else _lib._dev_f32(imgs.to(torch.float32), "imgs")
                proj = _lib._dev_f32(proj_matrices.to(device), "proj_matrices")
                dv = _lib._dev_f32(depth_values.to(device), "depth_values")
                blob, fblob = self._weights_blob(device), self._feature_blob(device)
                dt = _lib.dtype_code(self.storage_dtype)
                ws = self._forward_workspace(device, N, H, W, D, dt)
                depth = torch.empty((B, H // 4, W // 4), dtype=torch.float32, device=device)
                conf = torch.empty((B, H // 4, W // 4), dtype=torch.float32, device=device)
                # steps 1-4 (reference mvsnet.py:125-218): one enqueue per batch item, same stream
                for b in range(B):
                    _lib.forward_images(imgs_f[b], proj[b], dv[b], fblob, blob, ws, depth[b], conf[b], dtype=dt)
            return {"depth": depth, "photometric_confidence": conf}
        with torch.cuda.device(device), torch.no_grad():
            if imgs.dtype == torch.uint8:   # the torch FeatureNet path: the loader's conversion, on the device
                imgs = (imgs.permute(0, 1, 4, 2, 3) if u8_hwc else imgs).to(torch.float32) / 255.0
            # step 1. feature extraction (reference mvsnet.py:125), all B*N images in one call
            feats = self.feature(imgs.reshape(B * N, imgs.shape[2], H, W).to(torch.float32))
            C, h, w = feats.shape[1], feats.shape[2], feats.shape[3]
            feats = feats.reshape(B, N, C, h, w).contiguous()
            proj = _lib._dev_f32(proj_matrices.to(device), "proj_matrices")
            dv = _lib._dev_f32(depth_values.to(device), "depth_values")
            blob = self._weights_blob(device)
            dt = _lib.dtype_code(self.storage_dtype)
            ws = self._workspace(device, N, D, h, w, dt)
            depth = torch.empty((B, h, w), dtype=torch.float32, device=device)
            conf = torch.empty((B, h, w), dtype=torch.float32, device=device)
            # steps 2-4 (reference mvsnet.py:145-218): one enqueue per batch item, same stream
            for b in range(B):
                _lib.depth_infer(feats[b], proj[b], dv[b], blob, ws, depth[b], conf[b], dtype=dt)
        return {"depth": depth, "photometric_confidence": conf}
