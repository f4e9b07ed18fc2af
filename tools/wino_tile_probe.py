#!/usr/bin/env python3
"""GPU-box probe: time of one Winograd F(4x4,3x3) conv (transforms + 36 batched GEMMs) per layer shape and GEMM tile; -1 = heuristic."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")
SHAPES = [("layer4 512->512 d4", 128, 256, 512, 512, 4), ("layer4 256->512 d4", 128, 256, 256, 512, 4), ("layer3 256->256 d2", 128, 256, 256, 256, 2),
          ("layer3.0 128->256 d1", 128, 256, 128, 256, 1), ("head 512->128", 128, 256, 512, 128, 1), ("layer2 128->128", 128, 256, 128, 128, 1),
          ("layer1 64->64", 256, 512, 64, 64, 1), ("native l4 512->512 d4", 97, 193, 512, 512, 4), ("native l3 256->256 d2", 97, 193, 256, 256, 2)]
names = {3: "128x128", 4: "64x128", 5: "128x64", -1: "auto"}
o = lib.opts(winograd=4)
for (nm, H, W, Cin, Cout, d) in SHAPES:
    row = []
    for t in (-1, 3, 4, 5):
        ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, 3, 1, d, t, 20, ctypes.byref(o), None) for _ in range(2))
        row.append("%s %.3f" % (names[t], ms))
    print("%-30s " % nm + " | ".join(row), flush=True)
