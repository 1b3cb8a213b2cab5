"""MI355X-native RANSAC essential-matrix / pose-recovery / triangulation hot path.

Host code is Python and mirrors the reference's ``lib/ransac`` and ``lib/epipolar`` call signatures;
all numeric work of the path runs in hand-written gfx950 HIP kernels behind the C ABI declared in
``include/sfm_hip.h`` (``csrc/libsfm_hip.so``).  There is no CPU fallback: without the library or a
GPU the numeric entry points raise.
"""
__version__ = "0.1.0"
