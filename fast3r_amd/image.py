"""`load_images` with the reference's signature and output (fast3r/dust3r/utils/image.py:76-159), resampling on the GPU
(SURVEY.md section 8f, rank 3).

What stays on the host: file listing, PIL decode, EXIF transpose, the optional lossless rotate / crop-to-landscape (:106-130) -- byte
shuffles PIL does in microseconds -- and the filter tables.  What moves to the GPU: the antialiased resize (PIL LANCZOS when shrinking,
BICUBIC otherwise, :68-74), the centre crop to multiples of 16 (:141-150) and `ImgNorm` (:32), i.e. everything that touches every
pixel with arithmetic; the decoded bytes go up once (uint8, 3 B / pixel) and the network input comes out in place, on the device, as
(1, 3, H, W) fp32.  Bit-exact with PIL + torchvision (tests/test_image.py): Pillow's resize is fixed-point integer work
(src/libImaging/Resample.c) and the tables below are computed exactly as Pillow computes them.
No torchvision / cv2 dependency (the reference imports both at module import).
"""
import math
import os
from functools import lru_cache

import numpy as np
import torch

from . import _lib
from ._lib import F3RError, check, ptr, stream_ptr

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


_FILTERS = {"lanczos": (_lanczos, 3.0), "bicubic": (_bicubic, 2.0)}


@lru_cache(maxsize=256)
def resample_tables(in_size, out_size, filter_name):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (double precision, round half away from zero into 22-bit fixed point):
    (ksize, bounds int32 (out, 2) = [first source index, count], kk int32 (out, ksize)).  Cached: a scene's images share their size."""
    fn, support0 = _FILTERS[filter_name]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _device_tables(in_size, out_size, filter_name, dev, cache):
    key = (in_size, out_size, filter_name, str(dev))
    if key not in cache:
        ksize, b, k = resample_tables(in_size, out_size, filter_name)
        cache[key] = (ksize, torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev))
    return cache[key]


def resize_u8(img_u8, new_w, new_h, filter_name, _cache=None):
    """PIL.Image.resize((new_w, new_h), LANCZOS | BICUBIC) of an (H, W, 3) uint8 CUDA tensor: horizontal pass, then vertical pass."""
    if img_u8.device.type != "cuda":
        raise F3RError(f"fast3r_amd.image.resize_u8 runs on the ROCm GPU (image is on {img_u8.device}); there is no CPU fallback")
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3
    cache = {} if _cache is None else _cache
    dev = img_u8.device
    cur = img_u8.contiguous()
    H, W, _ = cur.shape
    if new_w != W:
        ksize, b, k = _device_tables(W, new_w, filter_name, dev, cache)
        out = torch.empty((H, new_w, 3), dtype=torch.uint8, device=dev)
        check(_lib.lib().f3r_resample_u8(ptr(cur), ptr(out), H, W, 1, new_w, ptr(b), ptr(k), ksize, stream_ptr()), "f3r_resample_u8")
        cur, W = out, new_w
    if new_h != H:
        ksize, b, k = _device_tables(H, new_h, filter_name, dev, cache)
        out = torch.empty((new_h, W, 3), dtype=torch.uint8, device=dev)
        check(_lib.lib().f3r_resample_u8(ptr(cur), ptr(out), H, W, 0, new_h, ptr(b), ptr(k), ksize, stream_ptr()), "f3r_resample_u8")
        cur = out
    return cur


def img_norm_crop(img_u8, box):
    """crop box (left, top, right, bottom) + ImgNorm: (H, W, 3) uint8 CUDA tensor -> (1, 3, h, w) fp32."""
    H, W, _ = img_u8.shape
    l, t, r, b = (int(v) for v in box)
    out = torch.empty((1, 3, b - t, r - l), dtype=torch.float32, device=img_u8.device)
    check(_lib.lib().f3r_imgnorm_u8(ptr(img_u8.contiguous()), ptr(out), H, W, l, t, r - l, b - t, stream_ptr()), "f3r_imgnorm_u8")
    return out


def _resized_size(size_wh, long_edge_size):
    """_resize_pil_image (:68-74): filter choice and the rounded new size."""
    S = max(size_wh)
    interp = "lanczos" if S > long_edge_size else "bicubic"
    return tuple(int(round(x * long_edge_size / S)) for x in size_wh), interp


def load_images(folder_or_list, size, square_ok=False, verbose=True, rotate_clockwise_90=False, crop_to_landscape=False, device="cuda"):
    """open and convert all images in a list or folder to the input format of Fast3R (reference :76-159); `img` tensors live on `device`."""
    import PIL.Image
    from PIL.ImageOps import exif_transpose
    if isinstance(folder_or_list, str):
        if verbose:
            print(f">> Loading images from {folder_or_list}")
        root, folder_content = folder_or_list, sorted(os.listdir(folder_or_list))
    elif isinstance(folder_or_list, list):
        if verbose:
            print(f">> Loading a list of {len(folder_or_list)} images")
        root, folder_content = "", folder_or_list
    else:
        raise ValueError(f"bad {folder_or_list=} ({type(folder_or_list)})")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise F3RError(f"fast3r_amd.load_images resamples on the ROCm GPU (device={device}); there is no CPU fallback")
    supported = (".jpg", ".jpeg", ".png")
    imgs, cache = [], {}
    for path in folder_content:
        if not path.lower().endswith(supported):
            continue
        img = exif_transpose(PIL.Image.open(os.path.join(root, path))).convert("RGB")
        if rotate_clockwise_90:
            img = img.rotate(-90, expand=True)
        if crop_to_landscape:  # :109-130
            desired = 4 / 3
            width, height = img.size
            if width / height > desired:
                new_width = int(height * desired)
                left = (width - new_width) // 2
                img = img.crop((left, 0, left + new_width, height))
            else:
                new_height = int(width / desired)
                top = (height - new_height) // 2
                img = img.crop((0, top, width, top + new_height))
        W1, H1 = img.size
        if size == 224:  # resize short side to 224 (then crop)
            (W, H), interp = _resized_size(img.size, round(size * max(W1 / H1, H1 / W1)))
        else:            # resize long side to `size`
            (W, H), interp = _resized_size(img.size, size)
        u8 = torch.from_numpy(np.array(img)).to(dev)  # (H1, W1, 3) uint8: the only upload
        u8 = resize_u8(u8, W, H, interp, cache)
        cx, cy = W // 2, H // 2
        if size == 224:
            half = min(cx, cy)
            box = (cx - half, cy - half, cx + half, cy + half)
        else:
            halfw, halfh = ((2 * cx) // 16) * 8, ((2 * cy) // 16) * 8
            if not square_ok and W == H:
                halfh = 3 * halfw / 4
            box = (cx - halfw, cy - halfh, cx + halfw, cy + halfh)
        tensor = img_norm_crop(u8, box)
        H2, W2 = tensor.shape[-2:]
        if verbose:
            print(f" - adding {path} with resolution {W1}x{H1} --> {W2}x{H2}")
        imgs.append(dict(img=tensor, true_shape=np.int32([[H2, W2]]), idx=len(imgs), instance=str(len(imgs))))
    assert imgs, "no images foud at " + root
    if verbose:
        print(f" (Found {len(imgs)} images)")
    return imgs
