#!/usr/bin/env python3
"""GPU-box: ResNet layer1's conv (64 -> 64, 3x3, on the 1/4-resolution map) on the exact-fp32 kernel and on the split kernel of precision 2
(td_conv_ad_b3.h, fusion bit 524288), alternating.    python tools/adb3_probe.py [HxW of the map] [iters] [both|fp32|split]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")
H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256x512").split("x"))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
which = sys.argv[3] if len(sys.argv) > 3 else "both"
base = lib.opts().fusion
for rnd in range(3):
    for tag, kw in (("fp32 ", {}), ("split", {"precision": 2, "fusion": base | 524288})):
        if which not in ("both", tag.strip()): continue
        o = lib.opts(**kw)
        ms = lib.tdnet_bench_conv(H, W, 64, 64, 3, 1, 1, -1, iters, ctypes.byref(o), None)
        print("%s %dx%d 64->64 k3: %.4f ms = %.1f TFLOP/s" % (tag, H, W, ms, 2.0 * H * W * 64 * 64 * 9 / ms / 1e9))
