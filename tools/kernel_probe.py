#!/usr/bin/env python3
"""GPU-box probe: sustained fp32-MFMA ceiling and the conv kernel variants on the real TDNet layer shapes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes  # noqa: E402
import torch  # noqa: E402

from tdnet_amd import _capi  # noqa: E402

lib = _capi.test_lib()
torch.zeros(1, device="cuda")
out = {"mfma_peak_tflops": {}}
for wps in (1, 2, 4):
    out["mfma_peak_tflops"][str(wps)] = round(lib.tdnet_bench_mfma_peak(wps, 4000, None), 2)
print(json.dumps(out), flush=True)
# (name, H, W, Cin, Cout, KS, stride, dil) at 1024x2048 input
SHAPES = [("layer4 512->512 d4", 128, 256, 512, 512, 3, 1, 4), ("layer4 256->512 d4", 128, 256, 256, 512, 3, 1, 4),
          ("layer3 256->256 d2", 128, 256, 256, 256, 3, 1, 2), ("layer2 128->128", 128, 256, 128, 128, 3, 1, 1),
          ("layer1 64->64", 256, 512, 64, 64, 3, 1, 1), ("head 512->128", 128, 256, 512, 128, 3, 1, 1),
          ("enc_v 1x1 512->512", 128, 256, 512, 512, 1, 1, 1), ("ds 1x1 256->512", 128, 256, 256, 512, 1, 1, 1)]
names = ["128x128", "64x128", "128x64", "128x128D", "64x128D", "128x64D"]
# quantisation probe: the same layer4 conv at 1, 2, 4, 8 workgroup rounds (M = 16k .. 131k pixels)
for Hq in (64, 128, 256, 512):
    gf = 2.0 * Hq * 256 * 512 * 512 * 9 / 1e9
    ms = lib.tdnet_bench_conv(Hq, 256, 512, 512, 3, 1, 4, 3, 10, ctypes.byref(lib.opts(winograd=0)), None)
    print("layer4 512->512 d4 at %dx256 (%d blocks): %.3f ms %.1f TF" % (Hq, Hq * 256 // 128 * 4, ms, gf / ms), flush=True)
for prec in (0, 1):
  o = lib.opts(winograd=0, precision=prec)          # this probe measures the direct kernels
  print("---- conv precision:", "fp16-input MFMA" if prec else "fp32 MFMA", flush=True)
  for (nm, H, W, Cin, Cout, KS, st, dil) in SHAPES:
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    gf = 2.0 * Ho * Wo * Cout * Cin * KS * KS / 1e9
    row = {}
    for t in (range(6) if prec == 0 else (3, 4, 5)):
        if Cout <= 64 and t not in (2, 5):
            continue
        ms = lib.tdnet_bench_conv(H, W, Cin, Cout, KS, st, dil, t, 20, ctypes.byref(o), None)
        row[names[t]] = "%.3f ms %.1f TF" % (ms, gf / ms)
    print("%-22s %6.1f GFLOP  " % (nm, gf) + "  ".join("%s: %s" % kv for kv in row.items()), flush=True)
