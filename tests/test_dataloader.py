"""cityscapesLoader keeps the reference's contract (Testing/dataloader.py:44-88): sorted PNG glob, item =
[img[1,3,H,W] fp32, name, folder, (W,H)], ImageNet normalisation in float64 -> float32, 19-colour palette.

PARITY UNPINNED for one function: `resize_linear_u8` restates cv2.resize's INTER_LINEAR (OpenCV's 11-bit fixed-point coefficients,
half-pixel centres; Testing/dataloader.py:64) and is pinned here by hand-derived answers only -- neither this image nor the GPU box
has cv2, so no cv2-generated fixture exists.  The day an image with cv2 is available: run cv2.resize on the arrays of
test_resize_is_cv2_inter_linear_not_pil_bilinear below, commit input/output pairs under tests/golden/ with the generating script, and compare
bit for bit.  The headline metric does not depend on it (synthetic tensors)."""
import os

import numpy as np
import torch

from tdnet_amd.dataloader import cityscapesLoader, resize_linear_u8


def test_loader_contract(tmp_path):
    from PIL import Image
    d = tmp_path / "vid1"
    d.mkdir()
    rng = np.random.default_rng(0)
    imgs = {}
    for name in ("b_000002.png", "a_000001.png"):
        a = rng.integers(0, 256, (20, 40, 3), dtype=np.uint8)
        Image.fromarray(a).save(d / name)
        imgs[name] = a
    ld = cityscapesLoader(img_path=str(d), in_size=(20, 40))          # same size -> resize is the identity
    assert ld.files_num == 2 and [os.path.basename(f) for f in ld.files] == ["a_000001.png", "b_000002.png"]
    ld.load_frames()
    img, name, folder, size = ld.data[0]
    assert name == "a_000001.png" and folder == "vid1" and size == (40, 20)      # 4th field is the TARGET (W,H): dataloader.py:73
    assert img.shape == (1, 3, 20, 40) and img.dtype == torch.float32
    exp = ((imgs[name] / 255.0 - np.array([.485, .456, .406])) / np.array([.229, .224, .225])).transpose(2, 0, 1)[None]
    assert np.abs(img.numpy() - exp.astype(np.float32)).max() < 1e-6
    ld2 = cityscapesLoader(img_path=str(d), in_size=(10, 24))
    ld2.load_frames()
    assert ld2.data[1][0].shape == (1, 3, 10, 24)


def test_resize_is_cv2_inter_linear_not_pil_bilinear():
    """Testing/dataloader.py:64 `cv2.resize(img, self.size)`: INTER_LINEAR, half-pixel centres, border clamp, no antialiasing,
    11-bit fixed point.  cv2 is not installed here, so the pins are known answers derived by hand from that definition, the float
    formula, and exact identities."""
    import torch.nn.functional as F
    # 1-D known answers: 2 -> 4 samples sit at source positions -0.25 (clamped), 0.25, 0.75, 1.25 (clamped)
    row = np.array([[[0], [100]]], np.uint8)                                       # H=1, W=2
    assert resize_linear_u8(row, (4, 1))[0, :, 0].tolist() == [0, 25, 75, 100]
    col = np.array([[[10]], [[250]]], np.uint8)                                     # H=2, W=1
    assert resize_linear_u8(col, (1, 4))[:, 0, 0].tolist() == [10, 70, 190, 250]
    # 4 -> 2: positions 0.5 and 2.5 -> plain means of neighbours (1 + 2) / 2 etc.; exact 2x downscale = 2x2 box average, rounded half up
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    box = (a[0::2, 0::2].astype(np.int64) + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert np.array_equal(resize_linear_u8(a, (48, 32)), box.astype(np.uint8))
    assert np.array_equal(resize_linear_u8(a, (96, 64)), a)                         # same size: a copy
    # generic scales (the harness's 1024x2048 -> 769x1537 is a 0.75x shrink): within one grey level of the float formula
    # (torch bilinear, align_corners=False, antialias=False = the same geometry), and NOT PIL's antialiased BILINEAR
    for (hd, wd) in ((48, 72), (37, 59), (100, 131), (64, 33)):
        got = resize_linear_u8(a, (wd, hd)).astype(np.float64)
        ref = F.interpolate(torch.from_numpy(a).permute(2, 0, 1)[None].double(), (hd, wd), mode="bilinear", align_corners=False,
                            antialias=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - ref).max() <= 1.0, (hd, wd, np.abs(got - ref).max())
        assert np.abs(got - ref).mean() <= 0.3
    from PIL import Image
    pil = np.asarray(Image.fromarray(a).resize((33, 21), Image.BILINEAR)).astype(np.float64)
    mine = resize_linear_u8(a, (33, 21)).astype(np.float64)
    assert np.abs(pil - mine).max() > 8                                            # a ~3x shrink of noise: antialiasing changes the picture


def test_palette():
    ld = cityscapesLoader(img_path="/nonexistent", in_size=(8, 8))
    lab = np.arange(19).reshape(1, 19)
    rgb = ld.decode_segmap(lab)
    assert rgb.shape == (1, 19, 3)
    assert rgb[0, 0].tolist() == [128, 64, 128] and rgb[0, 13].tolist() == [0, 0, 142] and rgb[0, 18].tolist() == [119, 11, 32]
