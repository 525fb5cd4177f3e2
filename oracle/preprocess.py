"""Oracle: detector pre-processing (letterbox), numpy.  TEST INFRASTRUCTURE ONLY.

Restates what `self.yolo.predict(image)` (/root/reference/src/circuit_analyzer.py:268) does before
the network: ultralytics `LetterBox(640, auto=True, stride=32)` built on `cv2.resize(INTER_LINEAR)`
and `cv2.copyMakeBorder(value=114)`, then channel flip, HWC->CHW, /255.
Neither ultralytics nor OpenCV is vendored or installed -> parity unpinned; the bilinear resize
follows OpenCV's published 8-bit fixed-point path (11-bit coefficients, two-pass, the
`((b*(S>>4))>>16 ... +2)>>2` vertical rounding).
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _axis_table(dst, src):
    scale = src / dst                                    # double, as cv::resize's inv scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0
    s[hi] = src - 1
    a1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int64)          # saturate_cast<short>(cvRound)
    a0 = np.rint((np.float32(1) - f) * np.float32(COEF_SCALE)).astype(np.int64)
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, a0, a1


def resize_linear_u8(img, dst_w, dst_h):
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC."""
    h, w = img.shape[:2]
    if (w, h) == (dst_w, dst_h):
        return img.copy()
    x0, x1, ax0, ax1 = _axis_table(dst_w, w)
    y0, y1, ay0, ay1 = _axis_table(dst_h, h)
    src = img.astype(np.int64)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]      # horizontal pass (int)
    s0, s1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (s0 >> 4)) >> 16) + ((ay1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(h, w, new_shape=640, stride=32, auto=True):
    r = min(new_shape / h, new_shape / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = new_shape - nw, new_shape - nh
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return nw, nh, top, bottom, left, right


def letterbox(img, new_shape=640, stride=32, auto=True, pad_value=114):
    h, w = img.shape[:2]
    nw, nh, top, bottom, left, right = letterbox_geometry(h, w, new_shape, stride, auto)
    if (w, h) != (nw, nh):
        img = resize_linear_u8(img, nw, nh)
    out = np.full((nh + top + bottom, nw + left + right, img.shape[2]), pad_value, dtype=np.uint8)
    out[top:top + nh, left:left + nw] = img
    return out


def yolo_preprocess(img, new_shape=640):
    """u8 HxWx3 -> f32 1x3xhxw: letterbox, reverse channel order, CHW, /255."""
    lb = letterbox(img, new_shape)
    chw = np.ascontiguousarray(lb[..., ::-1].transpose(2, 0, 1))
    return (chw.astype(np.float32) / 255.0)[None]
