"""Host-side mirror of the reference's models/module.py for the depth-inference path.

Same names, argument meaning and error behaviour as /root/reference/models/module.py:
  ConvBnReLU, ConvBnReLU3D  -- parameter containers (fix the checkpoint key names)
  homo_warping(src_fea, src_proj, ref_proj, depth_values)   module.py:96-139
  depth_regression(p, depth_values)                          module.py:144-147
The two functions run hand-written HIP kernels through the C ABI (include/mvs_abi.h); they
need CUDA(ROCm) float32 tensors and raise otherwise -- there is no CPU fallback.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


class ConvBnReLU(nn.Module):
    """2D conv + BN + ReLU block of FeatureNet (reference models/module.py:6-13)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad,
                              bias=False)
        self.bn = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class ConvBnReLU3D(nn.Module):
    """Parameter container for a CostRegNet block (reference models/module.py:26-33).

    The arithmetic runs in the HIP path (BN folded by mvs_pack_weights); this module only owns
    the parameters/buffers so that state_dict keys match the reference checkpoints.
    """

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad,
                              bias=False)
        self.bn = nn.BatchNorm3d(out_channels)

    def forward(self, x):  # pragma: no cover - never called on the product path
        raise RuntimeError("ConvBnReLU3D is a parameter container; CostRegNet runs in libmvs_hip")


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    """Differentiable-homography warp, inference only (reference models/module.py:96-139).

    src_fea [B,C,H,W], src_proj/ref_proj [B,4,4], depth_values [B,D] -> [B,C,D,H,W] (float32).
    """
    src_fea = _lib._dev_f32(src_fea, "src_fea")
    batch, channels, height, width = src_fea.shape
    num_depth = depth_values.shape[1]
    out = torch.empty((batch, channels, num_depth, height, width), dtype=torch.float32,
                      device=src_fea.device)
    lib = _lib.load()
    stream = _lib._stream(src_fea.device)
    with torch.cuda.device(src_fea.device):
        for b in range(batch):
            proj = torch.stack((ref_proj[b], src_proj[b])).to(torch.float32)
            rt = _lib.relative_proj(proj)
            dv = _lib._dev_f32(depth_values[b], "depth_values")
            _lib.check(lib.mvs_homo_warp(src_fea[b].data_ptr(), rt.data_ptr(), dv.data_ptr(),
                                         out[b].data_ptr(), channels, num_depth, height, width,
                                         stream))
    return out


def depth_regression(p, depth_values):
    """depth = sum_d p[:, d] * depth_values[:, d]  (reference models/module.py:144-147).

    p [B,D,H,W]; depth_values [B,D] (or [D], broadcast over the batch as the reference's
    `depth_values.view(*shape, 1, 1)` does for a 1-D tensor).
    """
    p = _lib._dev_f32(p, "p")
    batch, num_depth, height, width = p.shape
    dvs = _lib._dev_f32(depth_values.to(p.device), "depth_values")
    out = torch.empty((batch, height, width), dtype=torch.float32, device=p.device)
    lib = _lib.load()
    stream = _lib._stream(p.device)
    with torch.cuda.device(p.device):
        for b in range(batch):
            dv = dvs if dvs.dim() == 1 else dvs[b]
            _lib.check(lib.mvs_depth_regression(p[b].data_ptr(), dv.data_ptr(), out[b].data_ptr(),
                                                num_depth, height, width, stream))
    return out
