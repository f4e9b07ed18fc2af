#!/usr/bin/env python3
"""GPU-box probe: does the Winograd domain of a layer-4 conv run faster when producer -> consumer fit the 256 MB
Infinity Cache?  One 512 -> 512 conv at 128 x 256 (dilation 4: V and M are 151 MB each) through tdnet_bench_conv, on ONE stream:
    whole    the conv as one chunk           in -> 36 GEMMs -> out     V + M = 302 MB in flight
    half     two row classes (rows mod 2)    twice (in -> GEMMs -> out) on half the tiles, 151 MB per class
    quarter  four row classes (rows mod 4)   four times ...            76 MB per class (tdnet_opts.overlap bit 64)
The same three kernels (k_wino4_in_c<4>, k_gemm_dma, k_wino4_out_c<4>) in all three forms; same bytes, same FLOP, only the working set
between a producer and its consumer changes.  Run once plainly (ms per conv) and once per variant under
`rocprofv3 --kernel-trace --stats` for the per-kernel split:
    python tools/wino_l3_probe.py                       # all three, ms per conv
    python tools/wino_l3_probe.py --variant quarter     # one form (for a profiler run)"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="all", choices=["all", "whole", "half", "quarter"])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dil", type=int, default=4)
    ap.add_argument("--channels", type=int, default=512)
    a = ap.parse_args()
    import torch
    from tdnet_amd import _capi
    lib = _capi.test_lib()
    torch.zeros(1, device="cuda")
    forms = {"whole": 2 | 8 | 32, "half": 1 | 8 | 32, "quarter": 1 | 8 | 32 | 64}
    H, W, C = 128, 256, a.channels
    for name, ov in forms.items():
        if a.variant not in ("all", name):
            continue
        o = lib.opts(overlap=ov)
        best = min(lib.tdnet_bench_conv(H, W, C, C, 3, 1, a.dil, -1, a.iters, ctypes.byref(o), None) for _ in range(3))
        tiles = (H // 4) * (W // 4)
        gb = 36 * tiles * C * 4 * 2 * 2 / 1e9          # V written + read, M written + read
        print("%-8s overlap=%-3d %.3f ms per conv   (Winograd-domain traffic %.2f GB -> %.1f TB/s if it all went to HBM; GEMM FLOP %.1f G)"
              % (name, ov, best, gb, gb / best, 2.0 * 36 * tiles * C * C / 1e9), flush=True)


if __name__ == "__main__":
    main()
