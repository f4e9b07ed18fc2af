#!/usr/bin/env python3
"""GPU-box helper for counter passes: one conv shape, one kernel form of tdnet_opts.precision = 2, a few launches.
    python tools/b3_one.py [tile] [Cin] [Cout] [dil] [iters]      tile: 3 = loader waves (256 rows), 5 = matrix-only, 4 = 128-row tiles, -2 = the fp32 kernels"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 5
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dil = int(sys.argv[4]) if len(sys.argv) > 4 else 4
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
grid = int(sys.argv[6]) if len(sys.argv) > 6 else 1      # > 1: the persistent GEMM grid forced to that many workgroups (tdnet_opts.gemm_persistent)
o = lib.opts(overlap=40, gemm_persistent=grid) if tile == -2 else lib.opts(precision=2, overlap=40, gemm_persistent=grid)
ms = lib.tdnet_bench_conv(128, 256, Cin, Cout, 3, 1, dil, -1 if tile == -2 else tile, iters, ctypes.byref(o), None)
print("tile %d %d->%d d%d grid %d: %.4f ms per conv" % (tile, Cin, Cout, dil, grid, ms))
