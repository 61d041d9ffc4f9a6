"""Drop-in for the reference's `simple_knn` package (submodules/simple-knn): `from simple_knn._C import distCUDA2`
(scene/gaussian_model.py:20) resolves to the MI355X implementation in ibgs_amd/csrc/knn.hip."""
from . import _C  # noqa: F401
