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


# ------------------------------------------------------------------------------------------------------ input pipeline (host side)
class Geometry:
    """Everything `load_images` does to ONE picture, as numbers (no pixels touched): the lossless pre-crop on the decoded picture, the
    antialiased resize, the final centre crop.  The arithmetic is the reference's input contract (dust3r/utils/image.py:109-150):
    a different rounding anywhere changes the network's input size."""
    __slots__ = ("pre_crop", "resized", "filter", "box", "decoded")

    def __init__(self, decoded_wh, size, square_ok, crop_to_landscape):
        w, h = decoded_wh
        self.decoded = (w, h)
        self.pre_crop = None
        if crop_to_landscape:  # 4:3 window, centred (:109-130)
            if w / h > 4 / 3:
                keep = int(h * (4 / 3))
                left = (w - keep) // 2
                self.pre_crop = (left, 0, left + keep, h)
            else:
                keep = int(w / (4 / 3))
                top = (h - keep) // 2
                self.pre_crop = (0, top, w, top + keep)
            w, h = self.pre_crop[2] - self.pre_crop[0], self.pre_crop[3] - self.pre_crop[1]
        # 224: the SHORT side becomes 224, then a centred square; otherwise the LONG side becomes `size` (:132-138)
        long_edge = round(size * max(w / h, h / w)) if size == 224 else size
        self.resized, self.filter = _resized_size((w, h), long_edge)
        rw, rh = self.resized
        cx, cy = rw // 2, rh // 2
        if size == 224:
            half = min(cx, cy)
            self.box = (cx - half, cy - half, cx + half, cy + half)
        else:  # both sides down to multiples of 16 around the centre; a square picture becomes 4:3 unless square_ok (:144-148)
            hw, hh = ((2 * cx) // 16) * 8, ((2 * cy) // 16) * 8
            if not square_ok and rw == rh:
                hh = 3 * hw / 4
            self.box = (cx - hw, cy - hh, cx + hw, cy + hh)

    @property
    def out_hw(self):
        l, t, r, b = (int(v) for v in self.box)
        return b - t, r - l


def _image_extensions():
    """.jpg / .jpeg / .png, plus .heic / .heif when pillow_heif can be imported (the reference's optional dependency, image.py:24-30)."""
    exts = [".jpg", ".jpeg", ".png"]
    try:
        from pillow_heif import register_heif_opener
        register_heif_opener()
        exts += [".heic", ".heif"]
    except ImportError:
        pass
    return tuple(exts)


def _decode(path, rotate_clockwise_90):
    """File -> upright RGB PIL image (EXIF orientation applied, optional lossless quarter turn): host work, microseconds of byte shuffling."""
    import PIL.Image
    from PIL.ImageOps import exif_transpose
    pic = exif_transpose(PIL.Image.open(path)).convert("RGB")
    return pic.rotate(-90, expand=True) if rotate_clockwise_90 else pic


def load_images(folder_or_list, size, square_ok=False, verbose=True, rotate_clockwise_90=False, crop_to_landscape=False, device="cuda"):
    """Open and convert all images of a folder or list to the input format of Fast3R -- same arguments, ordering, filtering by extension,
    errors and per-image dict (`img` (1,3,H,W) in [-1,1], `true_shape` int32 [[H,W]], `idx`, `instance`) as the reference
    (fast3r/dust3r/utils/image.py:76-159); the `img` tensors are produced on `device` (a ROCm GPU) by the resampling kernels."""
    if isinstance(folder_or_list, str):
        root, names = folder_or_list, sorted(os.listdir(folder_or_list))
        if verbose:
            print(f">> Loading images from {root}")
    elif isinstance(folder_or_list, list):
        root, names = "", folder_or_list
        if verbose:
            print(f">> Loading a list of {len(names)} images")
    else:
        raise ValueError(f"bad {folder_or_list=} ({type(folder_or_list)})")  # :91
    dev = torch.device(device)
    if dev.type != "cuda":
        raise F3RError(f"fast3r_amd.load_images resamples on the ROCm GPU (device={device}); there is no CPU fallback")
    exts = _image_extensions()
    table_cache, out = {}, []
    with torch.cuda.device(dev):
        for name in names:
            if not name.lower().endswith(exts):
                continue
            pic = _decode(os.path.join(root, name), rotate_clockwise_90)
            geo = Geometry(pic.size, size, square_ok, crop_to_landscape)
            if geo.pre_crop is not None:
                pic = pic.crop(geo.pre_crop)
            pixels = torch.from_numpy(np.array(pic)).to(dev)  # (h, w, 3) uint8 (a writable copy of the decoder's buffer): the only upload of this picture
            pixels = resize_u8(pixels, geo.resized[0], geo.resized[1], geo.filter, table_cache)
            tensor = img_norm_crop(pixels, geo.box)
            h_out, w_out = tensor.shape[-2:]
            if verbose:
                print(f" - adding {name} with resolution {geo.decoded[0]}x{geo.decoded[1]} --> {w_out}x{h_out}")
            out.append(dict(img=tensor, true_shape=np.int32([[h_out, w_out]]), idx=len(out), instance=str(len(out))))
    assert out, "no images found at " + root  # the reference asserts here too (:156)
    if verbose:
        print(f" (Found {len(out)} images)")
    return out
