"""Input pipeline (reference fast3r/dust3r/utils/image.py:68-159).  The arithmetic of the reference is
Pillow's resize + torchvision's ToTensor / Normalize; Pillow is installed here, so parity is PINNED on PIL itself:
  CPU  the numpy restatement (oracle/resize_oracle.py) and the product's filter tables == PIL, bit for bit;
  GPU  the HIP resize == PIL bit for bit, and load_images == the reference recipe executed with PIL + fp32 numpy on files on disk."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import resize_oracle as RO

CASES = [(37, 53, 14, 20), (100, 75, 512, 384), (480, 640, 384, 512), (333, 500, 341, 512), (64, 64, 64, 48), (50, 50, 50, 50), (17, 9, 40, 21)]


def _img(H, W, seed):
    rng = np.random.default_rng(seed)
    base = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    base[: H // 3, : W // 2] = 255  # saturated / flat regions exercise the clip and the negative lobes
    base[H // 2:, W // 2:] = 0
    return base


def test_oracle_and_tables_equal_pil():
    from fast3r_amd.image import resample_tables
    for H, W, nh, nw in CASES:
        img = _img(H, W, H * W)
        for f, name in ((Image.LANCZOS, "lanczos"), (Image.BICUBIC, "bicubic")):
            ref = np.asarray(Image.fromarray(img).resize((nw, nh), f))
            assert np.array_equal(RO.resize_u8(img, nw, nh, name), ref), (H, W, nh, nw, name)
            for i, o in ((W, nw), (H, nh)):  # the product's (cached) tables are the restatement's tables
                k1, b1, c1 = RO.precompute_coeffs(i, o, name)
                k2, b2, c2 = resample_tables(i, o, name)
                assert k1 == k2 and np.array_equal(b1, b2) and np.array_equal(c1, c2)


def test_img_norm_restatement():
    u8 = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    x = RO.img_norm(u8)
    t = (torch.from_numpy(u8).permute(2, 0, 1).to(torch.float32).div(255) - 0.5) / 0.5  # torchvision's ToTensor + Normalize arithmetic
    assert np.array_equal(x, t.numpy())


def _reference_recipe(path, size, square_ok=False):
    """load_images for one file with PIL (image.py:105-158), ImgNorm in fp32 numpy."""
    from PIL.ImageOps import exif_transpose
    img = exif_transpose(Image.open(path)).convert("RGB")
    W1, H1 = img.size
    S = max(img.size)
    interp = Image.LANCZOS if S > size else Image.BICUBIC
    img = img.resize(tuple(int(round(x * size / S)) for x in img.size), interp)
    W, H = img.size
    cx, cy = W // 2, H // 2
    halfw, halfh = ((2 * cx) // 16) * 8, ((2 * cy) // 16) * 8
    if not square_ok and W == H:
        halfh = 3 * halfw / 4
    img = img.crop((cx - halfw, cy - halfh, cx + halfw, cy + halfh))
    return RO.img_norm(np.asarray(img))[None], np.int32([img.size[::-1]])


@pytest.mark.gpu
def test_hip_resize_equals_pil(built_lib):
    from fast3r_amd.image import resize_u8
    for H, W, nh, nw in CASES + [(1200, 1600, 384, 512)]:
        img = _img(H, W, 7 * H + W)
        for f, name in ((Image.LANCZOS, "lanczos"), (Image.BICUBIC, "bicubic")):
            ref = np.asarray(Image.fromarray(img).resize((nw, nh), f))
            got = resize_u8(torch.from_numpy(img).cuda(), nw, nh, name).cpu().numpy()
            assert np.array_equal(got, ref), (H, W, nh, nw, name, int(np.abs(got.astype(int) - ref.astype(int)).max()))


@pytest.mark.gpu
def test_load_images_equals_reference_recipe(built_lib, tmp_path):
    from fast3r_amd.image import load_images
    sizes = [(480, 640), (640, 480), (300, 300), (1000, 1500), (200, 320)]
    for i, (H, W) in enumerate(sizes):
        Image.fromarray(_img(H, W, i)).save(tmp_path / f"im{i:02d}.png")
    (tmp_path / "notes.txt").write_text("ignored")
    out = load_images(str(tmp_path), size=512, verbose=False)
    assert len(out) == len(sizes)
    for i, o in enumerate(out):
        ref, ts = _reference_recipe(str(tmp_path / f"im{i:02d}.png"), 512)
        assert o["img"].is_cuda and o["img"].dtype == torch.float32 and o["idx"] == i and o["instance"] == str(i)
        assert np.array_equal(o["true_shape"], ts) and o["true_shape"].dtype == np.int32
        assert np.array_equal(o["img"].cpu().numpy(), ref), i
        assert o["img"].shape[-1] % 16 == 0 and o["img"].shape[-2] % 16 == 0
    with pytest.raises(AssertionError):
        os.makedirs(tmp_path / "empty")
        load_images(str(tmp_path / "empty"), size=512, verbose=False)


def test_geometry_plan_equals_the_reference_arithmetic():
    """`Geometry` (fast3r_amd/image.py) = the numbers of load_images' per-picture transformations; checked against a line-by-line
    restatement of the reference's arithmetic (dust3r/utils/image.py:109-150) over sizes / modes incl. size=224 and crop_to_landscape."""
    import itertools
    from fast3r_amd.image import Geometry, _resized_size

    def reference(W1, H1, size, square_ok, crop):
        pre = None
        if crop:                                                       # :109-130
            if W1 / H1 > 4 / 3:
                nw = int(H1 * (4 / 3)); left = (W1 - nw) // 2; pre = (left, 0, left + nw, H1)
            else:
                nh = int(W1 / (4 / 3)); top = (H1 - nh) // 2; pre = (0, top, W1, top + nh)
            W1, H1 = pre[2] - pre[0], pre[3] - pre[1]
        (W, H), it = _resized_size((W1, H1), round(size * max(W1 / H1, H1 / W1)) if size == 224 else size)  # :132-138
        cx, cy = W // 2, H // 2
        if size == 224:
            half = min(cx, cy); box = (cx - half, cy - half, cx + half, cy + half)
        else:
            halfw, halfh = ((2 * cx) // 16) * 8, ((2 * cy) // 16) * 8
            if not square_ok and W == H:
                halfh = 3 * halfw / 4
            box = (cx - halfw, cy - halfh, cx + halfw, cy + halfh)     # :144-148
        return pre, (W, H), it, box

    sizes = [(4000, 3000), (3000, 4000), (1000, 1000), (640, 480), (517, 389), (233, 1000), (100, 100), (1920, 1080)]
    for (w, h), size, sq, crop in itertools.product(sizes, [224, 512, 384], [False, True], [False, True]):
        g = Geometry((w, h), size, sq, crop)
        assert (g.pre_crop, g.resized, g.filter, tuple(g.box)) == reference(w, h, size, sq, crop), (w, h, size, sq, crop)


# ------------------------------------------------------------------------------------------------ the reference's function, run for real
# tests/golden/load_images_cases.json (oracle/make_golden_images.py): fast3r.dust3r.utils.image.load_images imported from the reference
# checkout and run unmodified on deterministic PNG files (real Pillow; torchvision's ToTensor / Normalize through oracle/torchvision_stub.py):
# per returned view shape, true_shape, idx / instance and the SHA-256 of the float32 image.
def _load_images_fixture():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "load_images_cases.json")))


def test_recipe_restatement_equals_the_reference_function(tmp_path):
    """CPU: the PIL + numpy recipe the other tests use as their yardstick reproduces the reference function's bytes (size 512 calls)."""
    from oracle.make_golden_images import digest, write_pictures
    write_pictures(str(tmp_path))
    fix = _load_images_fixture()
    for case in fix["cases"]:
        kw = case["kwargs"]
        if set(kw) - {"size", "square_ok"} or kw["size"] == 224:
            continue
        for i, v in enumerate(case["views"]):
            ref, ts = _reference_recipe(str(tmp_path / f"im{i:02d}.png"), kw["size"], kw.get("square_ok", False))
            assert list(ref.shape) == v["shape"] and ts.tolist() == v["true_shape"]
            assert digest(torch.from_numpy(ref)) == v["sha256"], (kw, i)


@pytest.mark.gpu
def test_load_images_equals_the_reference_function_bit_for_bit(built_lib, tmp_path):
    from fast3r_amd.image import load_images
    from oracle.make_golden_images import digest, write_pictures
    write_pictures(str(tmp_path))
    fix = _load_images_fixture()
    for case in fix["cases"]:
        out = load_images(str(tmp_path), verbose=False, **case["kwargs"])
        assert len(out) == len(case["views"])
        for o, v in zip(out, case["views"]):
            assert list(o["img"].shape) == v["shape"] and np.asarray(o["true_shape"]).tolist() == v["true_shape"], (case["kwargs"], v)
            assert o["idx"] == v["idx"] and o["instance"] == v["instance"] and o["img"].dtype == torch.float32
            assert digest(o["img"]) == v["sha256"], (case["kwargs"], v["idx"])
