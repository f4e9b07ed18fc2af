#!/usr/bin/env python3
"""GPU-box probe: effect of de-phasing co-resident conv workgroups (tdnet_set_conv_stagger) per layer shape and on the frame."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import _capi
lib = _capi.lib(); torch.zeros(1, device="cuda"); lib.tdnet_set_conv_winograd(0)
SHAPES = [("layer4 512->512 d4", 128, 256, 512, 512, 3, 1, 4, 3), ("layer3 256->256 d2", 128, 256, 256, 256, 3, 1, 2, 3),
          ("layer2 128->128 (64x128)", 128, 256, 128, 128, 3, 1, 1, 4), ("layer1 64->64 (128x64)", 256, 512, 64, 64, 3, 1, 1, 5),
          ("enc_v 1x1 512->512", 128, 256, 512, 512, 1, 1, 1, 3)]
for st in (0, 1, 2, 3, 4, 6, 8):
    lib.tdnet_set_conv_stagger(st)
    row = []
    for (nm, H, W, Cin, Cout, KS, s, d, t) in SHAPES:
        gf = 2.0 * H * W * Cout * Cin * KS * KS / 1e9
        ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, KS, s, d, t, 20, None) for _ in range(2))
        row.append("%s %.1f TF" % (nm.split()[0], gf / ms))
    print("stagger %d: " % st + " | ".join(row), flush=True)
lib.tdnet_set_conv_stagger(0)
