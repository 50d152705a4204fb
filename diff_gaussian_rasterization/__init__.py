"""Drop-in module name for the reference's import
(`from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer`, models/modules/renderer/gaussian.py:9), served by the
MI355X-native implementation in gomavatar_amd.rasterizer."""
from gomavatar_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize"]
