"""repsurf_amd — MI355X (gfx950) native hot path of RepSurf-U.

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/repsurf_hip.h) -> lib/librepsurf_hip.so
  _lib.py          ctypes binding of that ABI (fails loudly when the library is missing)
  ops.py           torch-facing operators (allocation, streams, autograd) over the ABI
  classification/  drop-in mirror of the reference's `classification/` import tree
                   (modules.pointnet2_utils, modules.repsurface_utils, models.repsurf.*)
"""
__version__ = "0.1.0"
