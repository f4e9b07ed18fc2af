"""cityscapesLoader keeps the reference's contract (Testing/dataloader.py:44-88): sorted PNG glob, item =
[img[1,3,H,W] fp32, name, folder, (W,H)], ImageNet normalisation in float64 -> float32, 19-colour palette."""
import os

import numpy as np
import torch

from tdnet_amd.dataloader import cityscapesLoader


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


def test_palette():
    ld = cityscapesLoader(img_path="/nonexistent", in_size=(8, 8))
    lab = np.arange(19).reshape(1, 19)
    rgb = ld.decode_segmap(lab)
    assert rgb.shape == (1, 19, 3)
    assert rgb[0, 0].tolist() == [128, 64, 128] and rgb[0, 13].tolist() == [0, 0, 142] and rgb[0, 18].tolist() == [119, 11, 32]
