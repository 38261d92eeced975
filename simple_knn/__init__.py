"""Import shim so that `from simple_knn._C import distCUDA2` (/root/reference/scene/gaussian_model_ht.py:20)
resolves without the reference's un-vendored submodule (/root/reference/.gitmodules:1-3)."""
