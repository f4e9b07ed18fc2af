"""Winograd F(4x4,3x3) on the GPU (mode 3 = the default scope, 4 = every stride-1 3x3): operator parity at the real layer shapes and the
standard model gate (max|dlogit| <= 1e-3, tie-band label flips only) against the fp32 CPU oracle.  (F(2x2), modes 1 / 2, was removed in round 5.)"""
import pytest
import torch

import opcheck
from tdnet_amd import _capi

pytestmark = pytest.mark.gpu


def test_winograd_ops_and_model():
    lib, mem = _capi.test_lib(), opcheck.TorchMem()
    if True:
        import test_gpu_model as tm
        # F(4x4,3x3) on every stride-1 3x3
        for a in [(13, 21, 64, 128, 3, 1, 2, 1, True), (12, 30, 64, 160, 3, 1, 4, 1, False), (9, 17, 128, 256, 3, 1, 8, 0, True),
                  (5, 9, 256, 512, 3, 1, 16, 2, False), (1, 1, 32, 32, 3, 1, 1, 0, False), (97, 193, 256, 256, 3, 1, 2, 1, True)]:
            opcheck.conv(lib, mem, *a, tol=2e-4, opts={"winograd": 4})
        opcheck.conv(lib, mem, 128, 256, 512, 512, 3, 1, 4, 1, True, tol=5e-4, opts={"winograd": 4})    # K = 512 sums through the +-8 output transform
        opcheck.conv(lib, mem, 128, 256, 512, 512, 3, 1, 8, 1, True, tol=5e-4, opts={"winograd": 4})
        opcheck.conv(lib, mem, 128, 256, 256, 256, 3, 1, 2, 1, True, tol=5e-4, opts={"winograd": 4})
        tm._vs_oracle("td4", "resnet18", 257, 513, 8, kernel_opts={"winograd": 4})   # mode 4: stricter than the default (layer1 on F4 as well)
        tm._vs_oracle("td4", "resnet18", 1024, 2048, 5, kernel_opts={"winograd": 4})
