"""Drop-in package name for the reference's second native dependency: `from simple_knn._C import distCUDA2`
(reference scene/gaussian_model.py:21).  Implemented by libvegsrast.so (vegs_amd/csrc/knn.hip)."""
