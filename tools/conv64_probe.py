#!/usr/bin/env python3
"""GPU-box probe: the Cout = 64 convs (layer1 at 1/4 resolution) on k_conv_igemm<128,64> vs the A-direct kernel (fusion bit 32)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")
for (nm, H, W, Cin, Cout, KS, st, d) in [("layer1 64->64 @256x512", 256, 512, 64, 64, 3, 1, 1), ("layer1 64->64 @193x385", 193, 385, 64, 64, 3, 1, 1),
                                        ("deep stem 64->64 @512x1024", 512, 1024, 64, 64, 3, 1, 1)]:
    gf = 2.0 * (H // st) * (W // st) * Cout * Cin * KS * KS / 1e9
    row = []
    for fus in (0, 32):
        o = lib.opts(winograd=0, fusion=fus)
        ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, KS, st, d, -1, 20, ctypes.byref(o), None) for _ in range(3))
        row.append("%s %.3f ms %.1f TF" % ("A-direct" if fus else "LDS     ", ms, gf / ms))
    print("%-28s %6.1f GFLOP  %s" % (nm, gf, "   ".join(row)), flush=True)
