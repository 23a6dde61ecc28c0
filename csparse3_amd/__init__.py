"""MI355X-native sparse direct-solve backend behind SanPen/CSparse3's CSC conventions.

    from csparse3_amd.csc import CscMat, lusol, cholsol      # host mirror of the reference API
    from csparse3_amd import csc_hip                          # flat-array kernels + Factorization handle

Importing this package does not load the shared library; the first kernel call does.
"""
__version__ = "0.1.0"
