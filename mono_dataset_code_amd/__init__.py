"""MI355X-native photometric + FOV undistortion hot path of tum-vision/mono_dataset_code.

The product is two native libraries (see build.py):

  libmdc_hip.so   gfx950 HIP kernels behind the C ABI in include/mdc_hip.h
  libmdc_host.so  the reference's C++ classes (UndistorterFOV, PhotometricUndistorter,
                  ExposureImage) re-implemented on top of that ABI, plus a C facade

This Python package is plumbing only: ctypes bindings (capi), a synthetic
calibration / sequence writer (synth) and frame sharding helpers (shard) used by
tests/ and bench.py.  It never computes the per-frame maths itself and raises
if the native libraries are missing.
"""
from . import capi, synth, shard  # noqa: F401

__all__ = ["capi", "synth", "shard"]
