"""MI355X-native MVSNet depth-inference path (drop-in for the reference's `models` package).

    from scene_3dreconstruction_mvsnet_amd import MVSNet       # == models.MVSNet
    from scene_3dreconstruction_mvsnet_amd.module import homo_warping, depth_regression
"""
from .mvsnet import MVSNet  # noqa: F401

__all__ = ["MVSNet"]
