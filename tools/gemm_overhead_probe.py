#!/usr/bin/env python3
"""GPU-box probe: persistent 1x1 GEMM (k_gemm_persistent, ROLE 0) time vs K and vs tiles per workgroup -> per-launch and per-tile
fixed cost (t = launch + tiles_per_wg * (c + d K))."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")


o = lib.opts(winograd=0)

for tile in (3, 4):
    for H in (128, 512, 1024):                       # M = H * 256: 32768 (2 tiles of 128x128 per workgroup at N = 512), 131072, 262144
        row = []
        for K in (128, 256, 512, 1024, 2048):
            ms = min(lib.tdnet_bench_conv(H, 256, K, 512, 1, 1, 1, tile, 20, ctypes.byref(o), None) for _ in range(3))
            row.append("K=%d %.4f ms %.0f TF" % (K, ms, 2.0 * H * 256 * K * 512 / ms / 1e9))
        print("tile %s M=%6d N=512: %s" % ({3: "128x128", 4: "64x128"}[tile], H * 256, " | ".join(row)), flush=True)
