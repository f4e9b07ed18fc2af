"""Frame loader with the reference's API (Testing/dataloader.py:44-88): cityscapesLoader(img_path, in_size),
.load_frames(), .data = [[img[1,3,H,W] fp32, img_name, folder, (W,H)], ...], .decode_segmap(labels).

imageio / cv2 are not in this image.  PNGs are read with PIL; the resize is `resize_linear_u8`, a restatement of what
`cv2.resize(img, (W, H))` (dataloader.py:64: default INTER_LINEAR on a uint8 image) computes: half-pixel centres, NO antialiasing on
downscale (PIL's BILINEAR widens its support when shrinking, so it is not a stand-in), and OpenCV's 11-bit fixed-point arithmetic, so
the uint8 result is meant to be the one OpenCV's generic path produces.  cv2 itself is absent here, so the function is pinned by
hand-derived known answers, by the float bilinear formula to within one grey level and by the 2x-downscale = 2x2 box-average
identity (tests/test_dataloader.py), not by a cv2-generated fixture.  Normalisation follows dataloader.py:66-67 in float64.
"""
import os

import numpy as np
import torch


def recursive_glob(rootdir=".", suffix=""):
    return [os.path.join(looproot, filename) for looproot, _, filenames in os.walk(rootdir)
            for filename in filenames if filename.endswith(suffix)]


def _linear_coeffs(n_src, n_dst):
    """Source index and the two 11-bit weights per destination index, as OpenCV's resize builds them for INTER_LINEAR:
    fx = (float)((d + 0.5) * scale - 0.5), s = floor(fx), fx -= s, clamped at both borders; weights = round((1 - fx, fx) * 2048)."""
    scale = float(n_src) / float(n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    fx = (fx - s.astype(np.float32)).astype(np.float32)
    low = s < 0
    fx[low] = 0.0; s[low] = 0
    high = s >= n_src - 1
    fx[high] = 0.0; s[high] = n_src - 1
    w1 = np.rint(fx * np.float32(2048.0)).astype(np.int64)
    w0 = np.rint((np.float32(1.0) - fx) * np.float32(2048.0)).astype(np.int64)
    return s, np.minimum(s + 1, n_src - 1), w0, w1


def resize_linear_u8(img, size):
    """uint8 [H,W,C] -> uint8 [size[1], size[0], C]; size = (W, H) like cv2.resize's dsize (dataloader.py:52,64).
    Horizontal pass in int32 (pixel * 2048-scale weights), vertical pass with OpenCV's shifts:
        dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    Wd, Hd = int(size[0]), int(size[1])
    Hs, Ws = img.shape[:2]
    if (Hs, Ws) == (Hd, Wd):
        return img.copy()
    x0, x1, a0, a1 = _linear_coeffs(Ws, Wd)
    y0, y1, b0, b1 = _linear_coeffs(Hs, Hd)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]          # [Hs, Wd, C], <= 255 * 2048
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class cityscapesLoader():
    colors = [[128, 64, 128], [244, 35, 232], [70, 70, 70], [102, 102, 156], [190, 153, 153], [153, 153, 153],
              [250, 170, 30], [220, 220, 0], [107, 142, 35], [152, 251, 152], [0, 130, 180], [220, 20, 60],
              [255, 0, 0], [0, 0, 142], [0, 0, 70], [0, 60, 100], [0, 80, 100], [0, 0, 230], [119, 11, 32]]
    label_colours = dict(zip(range(19), colors))

    def __init__(self, img_path, in_size):
        self.img_path = img_path
        self.n_classes = 19
        self.files = sorted(recursive_glob(rootdir=self.img_path, suffix=".png"))
        self.files_num = len(self.files)
        self.data = []
        self.size = (in_size[1], in_size[0])            # (W, H) as dataloader.py:52
        self.mean = np.array([.485, .456, .406])
        self.std = np.array([.229, .224, .225])

    def normalise(self, img_u8):
        """uint8 HWC (already at self.size) -> fp32 [1,3,H,W], dataloader.py:66-71."""
        img = img_u8 / 255.0
        img = (img - self.mean) / self.std
        img = img.transpose(2, 0, 1)[np.newaxis, :]
        return torch.from_numpy(img).float()

    def load_frames(self):
        from PIL import Image
        for path in self.files:
            path = path.rstrip()
            img_name = path.split('/')[-1]
            folder = path.split('/')[-2]
            im = resize_linear_u8(np.asarray(Image.open(path).convert("RGB")), self.size)      # = cv2.resize(img, self.size), dataloader.py:64
            self.data.append([self.normalise(im), img_name, folder, self.size])

    def decode_segmap(self, temp):
        rgb = np.zeros((temp.shape[0], temp.shape[1], 3))
        for l in range(0, self.n_classes):
            rgb[temp == l] = self.label_colours[l]
        # labels outside 0..18 keep their raw value in all three channels, as dataloader.py:75-88 does
        other = (temp < 0) | (temp >= self.n_classes)
        rgb[other] = temp[other][:, None]
        return rgb
