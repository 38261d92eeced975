"""MI355X-native differentiable Gaussian rasterizer (drop-in for `diff_gaussian_rasterization`).

The directory name is not a Python identifier; import it with
`importlib.import_module("3dgs_hierarchical_training_amd")` or use the drop-in alias package
`diff_gaussian_rasterization` at the repo root.
"""
__version__ = "0.1.0"
