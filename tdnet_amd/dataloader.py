"""Frame loader with the reference's API (Testing/dataloader.py:44-88): cityscapesLoader(img_path, in_size),
.load_frames(), .data = [[img[1,3,H,W] fp32, img_name, folder, (W,H)], ...], .decode_segmap(labels).

imageio / cv2 are not in this image; PNGs are read and resized (bilinear) with PIL.  Pixel-exact cv2.resize parity is
not required for the measured path (synthetic tensors); normalisation follows dataloader.py:66-67 in float64.
"""
import os

import numpy as np
import torch


def recursive_glob(rootdir=".", suffix=""):
    return [os.path.join(looproot, filename) for looproot, _, filenames in os.walk(rootdir)
            for filename in filenames if filename.endswith(suffix)]


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
            im = Image.open(path).convert("RGB").resize(self.size, Image.BILINEAR)
            self.data.append([self.normalise(np.asarray(im)), img_name, folder, self.size])

    def decode_segmap(self, temp):
        rgb = np.zeros((temp.shape[0], temp.shape[1], 3))
        for l in range(0, self.n_classes):
            rgb[temp == l] = self.label_colours[l]
        # labels outside 0..18 keep their raw value in all three channels, as dataloader.py:75-88 does
        other = (temp < 0) | (temp >= self.n_classes)
        rgb[other] = temp[other][:, None]
        return rgb
