#!/usr/bin/env python3
"""GPU-box probe: direct vs Winograd F(4x4,3x3) time per layer shape (same launch path as the model).  (F(2x2) left the library in round 5;
its round-2/3 numbers are in profiles/r02*, r03*.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")
SHAPES = [("layer4 512->512 d4", 128, 256, 512, 512, 4), ("layer4 512->512 d8", 128, 256, 512, 512, 8), ("layer4 256->512 d4", 128, 256, 256, 512, 4),
          ("layer3 256->256 d2", 128, 256, 256, 256, 2), ("layer3.0 128->256 d1", 128, 256, 128, 256, 1), ("head 512->128", 128, 256, 512, 128, 1),
          ("layer2 128->128", 128, 256, 128, 128, 1), ("layer1 64->64", 256, 512, 64, 64, 1), ("native l4 512->512 d4", 97, 193, 512, 512, 4)]
for (nm, H, W, Cin, Cout, d) in SHAPES:
    gf = 2.0 * H * W * Cout * Cin * 9 / 1e9
    out = []
    for mode in (0, 4):
        o = lib.opts(winograd=mode)
        ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, 3, 1, d, -1, 20, ctypes.byref(o), None) for _ in range(2))
        out.append("%s %.3f ms (%.0f TF eff.)" % ({0: "direct", 4: "F4"}[mode], ms, gf / ms))
    print("%-24s %6.1f GFLOP  %s" % (nm, gf, "   ".join(out)), flush=True)
