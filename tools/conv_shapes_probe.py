#!/usr/bin/env python3
"""GPU-box probe: the direct-conv launches of the default frame (layer1, layer2.0's strided conv; stems are timed by kernel_probe),
isolated, best of three x 20 launches."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")
for (nm, H, W, Cin, Cout, KS, st, d) in [("layer1 64->64 3x3 @256x512", 256, 512, 64, 64, 3, 1, 1), ("layer2.0 64->128 3x3 s2 @256x512", 256, 512, 64, 128, 3, 2, 1),
                                        ("layer2.0 ds 64->128 1x1 s2", 256, 512, 64, 128, 1, 2, 1), ("layer4 512->512 3x3 d4 @128x256 (direct)", 128, 256, 512, 512, 3, 1, 4),
                                        ("layer3 256->256 3x3 d2 (direct)", 128, 256, 256, 256, 3, 1, 2)]:
    gf = 2.0 * (H // st) * (W // st) * Cout * Cin * KS * KS / 1e9
    row = []
    for prec in (0, 1):
        o = lib.opts(winograd=0, precision=prec)
        ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, KS, st, d, -1, 20, ctypes.byref(o), None) for _ in range(3))
        row.append("%s %.4f ms %.1f TF" % ("fp16" if prec else "fp32", ms, gf / ms))
    print("%-44s %6.1f GFLOP  %s" % (nm, gf, "   ".join(row)), flush=True)
