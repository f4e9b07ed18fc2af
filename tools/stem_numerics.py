#!/usr/bin/env python3
"""GPU box: is the packed-row 7x7 stem (fusion bit 65536) as accurate as the NHWC4 one?  (1) the stem operator alone against an fp64
evaluation, both kernels, He-initialised weights; (2) tests/test_gpu_model.py's reference-init stress (un-calibrated weights, td4-psp18
129x257, 5 frames) with either stem, direct and Winograd convs, over several clip seeds -- prints, does not gate."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opcheck                                              # noqa: E402
from tdnet_amd import _capi                                 # noqa: E402

ROWS = 65536


def stem_alone():
    lib, mem = _capi.test_lib(), opcheck.TorchMem()
    for H, W in ((129, 257), (257, 513), (300, 422)):
        g = np.random.default_rng(1)
        img = g.standard_normal((3, H, W)).astype(np.float32)
        w = (g.standard_normal((64, 3, 7, 7)) * (2.0 / (64 * 49)) ** 0.5).astype(np.float32)
        b = np.zeros(64, np.float32)
        ti, tw = torch.from_numpy(img)[None], torch.from_numpy(w)
        ref = F.max_pool2d(F.relu(F.conv2d(ti.double(), tw.double(), None, 2, 3)), 3, 2, 1)[0].permute(1, 2, 0).numpy()
        cpu = F.max_pool2d(F.relu(F.conv2d(ti, tw, None, 2, 3)), 3, 2, 1)[0].permute(1, 2, 0).double().numpy()
        print("stem %dx%d: cpu fp32 max %.3e rms %.3e" % (H, W, np.abs(cpu - ref).max(), ((cpu - ref) ** 2).mean() ** 0.5))
        for fu in (32, 32 | ROWS):
            di, out = mem.put(img), mem.empty(ref.shape)
            lib.check(lib.tdnet_op_stem(mem.ptr(di), H, W, w.ctypes.data, b.ctypes.data, ctypes.byref(lib.opts(fusion=fu)), mem.ptr(out), mem.stream))
            d = mem.get(out).astype(np.float64) - ref
            bad = np.argwhere(np.abs(d) > 1e-5)
            print("   fusion %6d: max %.3e rms %.3e; |d| > 1e-5 at %d places %s" % (fu, np.abs(d).max(), (d ** 2).mean() ** 0.5, len(bad), bad[:6].tolist()))


def stress():
    import test_gpu_model as tg
    default = _capi.test_lib().opts().fusion
    for seed in (1, 2, 3):
        for wino in (0, 3):
            for fu in (default & ~ROWS, default | ROWS):
                print("seed %d fusion %d:" % (seed, fu), end=" ")
                tg._reference_init_stress(129, 257, 5, wino, extra_opts={"fusion": fu}, seed=seed, gate=False)


if __name__ == "__main__":
    stem_alone()
    stress()
