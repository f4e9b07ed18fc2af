#!/usr/bin/env python3
"""GPU-box probe: the stem (layout change, 7x7 stride-2 conv, max-pool) at 1024x2048 through tdnet_op_stem; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel times (the op itself allocates and synchronises)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tdnet_amd import _capi
lib = _capi.test_lib()
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 0             # 2: with fusion bit 524288 the conv runs on the split kernel (td_conv_ad_b3.h)
H, W = 1024, 2048
img = torch.randn(3, H, W, device="cuda"); out = torch.empty(H // 4, W // 4, 64, device="cuda")
w = (np.random.default_rng(0).standard_normal((64, 3, 7, 7)) * 0.1).astype(np.float32); b = np.zeros(64, np.float32)
o = lib.opts(precision=prec, fusion=lib.opts().fusion | (524288 if prec >= 2 else 0))
for _ in range(12):
    lib.check(lib.tdnet_op_stem(img.data_ptr(), H, W, w.ctypes.data, b.ctypes.data, ctypes.byref(o), out.data_ptr(), None))
torch.cuda.synchronize()
