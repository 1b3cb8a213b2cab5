"""2-D cross-correlation (reference ``lib/common/correlate.py:4-39``), evaluated by ``correlate_kernel``."""
import numpy as np

from .. import _native, device
from .._native import check


def cross_correlate(image: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """Slide ``kernel`` over ``image`` with zero 'same' padding; only odd-sized square kernels.  The border of
    half a kernel width is left at zero (as the reference does); the result is float64, the size of ``image``."""
    if len(image.shape) != 2 or len(kernel.shape) != 2:
        raise ValueError("Only 2D single channel images are supported")
    if kernel.shape[0] != kernel.shape[1] or (kernel.shape[0] % 2) == 0:
        raise ValueError("Only odd-sized square kernels are supported")
    if image.shape[0] < kernel.shape[0] or image.shape[1] < kernel.shape[0]:
        raise ValueError("Kernel cannot be larger than image")
    return correlate_device(device.to_device(np.ascontiguousarray(image, dtype=np.float64)), kernel).cpu().numpy()


def correlate_device(image_t, kernel: np.ndarray):
    """Device-resident form: image tensor [H,W] f64 -> tensor [H,W] f64."""
    import torch

    lib = _native.load()
    kern = device.to_device(np.ascontiguousarray(kernel, dtype=np.float64))
    out = torch.empty_like(image_t)
    check(lib.sfm_cross_correlate(image_t.data_ptr(), image_t.shape[0], image_t.shape[1], kern.data_ptr(),
                                  int(kernel.shape[0]), out.data_ptr(), device._stream()), "sfm_cross_correlate")
    return out
