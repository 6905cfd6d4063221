"""TEST INFRASTRUCTURE -- generates tests/golden/load_images_cases.json by running the REAL reference's `load_images` on CPU.

    python -m oracle.make_golden_images          (build container only: needs /root/reference)

`fast3r.dust3r.utils.image.load_images` (dust3r/utils/image.py:76-159) is imported from the reference checkout and run unmodified on PNG
files written from deterministic synthetic pictures; Pillow is the real one, `torchvision.transforms` (not installed) resolves to
oracle/torchvision_stub.py (ToTensor / Normalize restated) and `cv2` to a dummy (the function does not use it).  Stored per call: the
option set and, per returned view, shape, true_shape, idx / instance and the SHA-256 of the float32 image bytes -- the product is compared
bit for bit (tests/test_image.py)."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "load_images_cases.json")

PICTURES = [(480, 640, 0), (640, 480, 1), (300, 300, 2), (1000, 1500, 3), (200, 320, 4), (517, 389, 5), (96, 96, 6)]  # (H, W, seed)
CALLS = [dict(size=512), dict(size=512, square_ok=True), dict(size=224), dict(size=384, rotate_clockwise_90=True), dict(size=512, crop_to_landscape=True)]


def synthetic_picture(H, W, seed):
    rng = np.random.default_rng(seed)
    base = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    base[: H // 3, : W // 2] = 255  # saturated / flat regions exercise the clip and the negative lobes of the filters
    base[H // 2:, W // 2:] = 0
    return base


def write_pictures(folder):
    for i, (H, W, seed) in enumerate(PICTURES):
        Image.fromarray(synthetic_picture(H, W, seed)).save(os.path.join(folder, f"im{i:02d}.png"))
    with open(os.path.join(folder, "notes.txt"), "w") as f:
        f.write("not a picture: skipped by extension (image.py:97-101)")


def digest(t):
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()


def main():
    from oracle import ref_loader, torchvision_stub
    ref_loader._STUB_ROOTS = tuple(r for r in ref_loader._STUB_ROOTS if r != "torchvision")
    torchvision_stub.install()
    ref_loader.install()
    from fast3r.dust3r.utils.image import load_images
    cases = []
    with tempfile.TemporaryDirectory() as d:
        write_pictures(d)
        for kw in CALLS:
            views = load_images(d, verbose=False, **kw)
            cases.append(dict(kwargs=kw, views=[dict(shape=list(v["img"].shape), true_shape=np.asarray(v["true_shape"]).tolist(), idx=v["idx"],
                                                     instance=v["instance"], sha256=digest(v["img"])) for v in views]))
            print(kw, [tuple(v["img"].shape[-2:]) for v in views])
    new = dict(pictures=[list(p) for p in PICTURES], cases=cases)
    if "--check" in sys.argv:  # compare with the committed fixture instead of writing it
        same = json.load(open(OUT)) == json.loads(json.dumps(new))
        print("load_images fixture:", "identical to the committed one" if same else "DIFFERS from the committed one")
        sys.exit(0 if same else 1)
    json.dump(new, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
