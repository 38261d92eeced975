"""Drop-in alias: `from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(/root/reference/scene/gaussian_model_ht.py:39-42, gaussian_renderer/__init__.py:12, scene/gaussian_model.py:37-40)
resolves to the MI355X-native implementation in 3dgs_hierarchical_training_amd/rasterizer.py.
Put the repo root on PYTHONPATH and the reference's trainer imports this package unmodified."""
import importlib as _importlib

_impl = _importlib.import_module("3dgs_hierarchical_training_amd.rasterizer")
GaussianRasterizationSettings = _impl.GaussianRasterizationSettings
GaussianRasterizer = _impl.GaussianRasterizer
rasterize_gaussians = _impl.rasterize_gaussians

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
