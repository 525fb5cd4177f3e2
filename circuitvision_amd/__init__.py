"""circuitvision_amd -- MI355X-native (gfx950) implementation of CircuitVision's dense-vision hot
path: the YOLO11 detector (+NMS) and the SAM 2.1 forward, behind the reference's own call
signatures (`YOLO(path).predict(img)`, `get_modified_sam2(...)(x)`, `SAM2Transforms`).

All arithmetic runs in hand-written HIP kernels reached through the C ABI in include/cvmi355.h;
there is no CPU or PyTorch-eager fallback (a missing library or GPU raises).
"""
from ._lib import CvmiError, F16, F32  # noqa: F401

__all__ = ["CvmiError", "F16", "F32", "YOLO"]


def __getattr__(name):
    if name == "YOLO":
        from .detector import YOLO
        return YOLO
    raise AttributeError(name)
