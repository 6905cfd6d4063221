"""TEST INFRASTRUCTURE ONLY -- a stand-in for the three torchvision transforms behind the reference's `ImgNorm`
(fast3r/dust3r/utils/image.py:32: Compose([ToTensor(), Normalize((0.5,)*3, (0.5,)*3)])); torchvision is not installed here.
Semantics restated from torchvision.transforms.functional: `to_tensor` of an 8-bit PIL image = HWC uint8 -> CHW float32 / 255;
`normalize` = (x - mean[:, None, None]) / std[:, None, None] in float32.  oracle/make_golden_images.py installs this module as
`torchvision` (and `.transforms`) so that the reference's own `load_images` runs unmodified around it (Pillow is real)."""
import sys
import types

import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    def __call__(self, pic):
        a = np.array(pic, copy=True)
        assert a.dtype == np.uint8 and a.ndim == 3
        return torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return t.clone().sub_(mean).div_(std)


def install():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tr.Compose, tr.ToTensor, tr.Normalize = Compose, ToTensor, Normalize
    tv.transforms = tr
    tv.__path__ = []
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr
