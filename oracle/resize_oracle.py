"""TEST INFRASTRUCTURE ONLY -- numpy restatement of Pillow's antialiased resize for 8-bit RGB images, the arithmetic behind the
reference's input pipeline (`load_images` -> `_resize_pil_image`, fast3r/dust3r/utils/image.py:68-74,76-159: PIL.Image.LANCZOS when
shrinking, PIL.Image.BICUBIC otherwise) and of `ImgNorm` (:32, torchvision ToTensor + Normalize(0.5, 0.5)).

Pillow is a third-party dependency of the reference (requirements: `pillow`; installed here: 12.2.0); its algorithm
(src/libImaging/Resample.c, unchanged since 3.x): per output coordinate a window of source pixels [xmin, xmax) and normalised filter
weights computed in double precision, converted to 22-bit fixed point with round-half-away; horizontal pass into an 8-bit
intermediate, then vertical pass; each pass accumulates from 1 << 21 and shifts right by 22 with saturation to [0, 255].
PINNED: tests/test_image.py compares this file with PIL itself, bit for bit, on random images and sizes; the HIP path (f3r_elem.hip) is
compared with PIL directly on the GPU box.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


FILTERS = {"lanczos": (_lanczos, 3.0), "bicubic": (_bicubic, 2.0)}


def precompute_coeffs(in_size, out_size, filter_name):
    """Resample.c::precompute_coeffs + normalize_coeffs_8bpc for box = (0, in_size): returns (ksize, bounds (out,2) int32 [xmin, count],
    kk (out, ksize) int32 fixed-point weights)."""
    fn, support0 = FILTERS[filter_name]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ww = 0.0
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = fn((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            v = k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, bounds, kk, axis):
    """one resampling pass along `axis` (0 = vertical, 1 = horizontal) of an (H, W, C) uint8 array."""
    src = img.astype(np.int64)
    out_size = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.empty(shape, dtype=np.uint8)
    for o in range(out_size):
        xmin, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        w = kk[o, :cnt].astype(np.int64)
        if axis == 1:
            acc = (src[:, xmin:xmin + cnt, :] * w[None, :, None]).sum(1)
        else:
            acc = (src[xmin:xmin + cnt, :, :] * w[:, None, None]).sum(0)
        acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
        res = np.clip(acc, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, o, :] = res
        else:
            out[o, :, :] = res
    return out


def resize_u8(img, new_w, new_h, filter_name):
    """PIL.Image.resize((new_w, new_h), LANCZOS | BICUBIC) of an (H, W, 3) uint8 array: horizontal pass, then vertical pass."""
    H, W, _ = img.shape
    out = img
    if new_w != W:
        _, b, k = precompute_coeffs(W, new_w, filter_name)
        out = _pass(out, b, k, axis=1)
    if new_h != H:
        _, b, k = precompute_coeffs(H, new_h, filter_name)
        out = _pass(out, b, k, axis=0)
    return out


def img_norm(u8_hwc):
    """ImgNorm (image.py:32): ToTensor (HWC uint8 -> CHW float32 / 255) then Normalize(0.5, 0.5): (x - 0.5) / 0.5, in fp32."""
    x = np.transpose(u8_hwc, (2, 0, 1)).astype(np.float32) / np.float32(255)
    return (x - np.float32(0.5)) / np.float32(0.5)
