#!/usr/bin/env python3
"""GPU-box probe of tdnet_opts.precision = 2 (fp32-accurate GEMMs on the bf16 MFMA, td_gemm_b3.h):
  1. error of one conv against an fp64 evaluation, exact-fp32 MFMA kernels vs the split kernels (same inputs);
  2. time of one Winograd F(4x4) conv (transforms + 36 GEMMs) per layer shape of the frame, both modes, whole conv and row-parity chunks.
Run it under `rocprofv3 --kernel-trace --stats` for the GEMM kernels' own durations."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from tdnet_amd import _capi
lib = _capi.test_lib(); torch.zeros(1, device="cuda")


def accuracy(H, W, Cin, Cout, KS, dil, seed=0):
    g = np.random.default_rng(seed)
    x = g.standard_normal((H, W, Cin)).astype(np.float32)
    w = (g.standard_normal((Cout, Cin, KS, KS)) / np.sqrt(Cin * KS * KS)).astype(np.float32)
    b = g.standard_normal(Cout).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double().permute(2, 0, 1)[None], torch.from_numpy(w).double(), torch.from_numpy(b).double(), 1, dil * (KS // 2), dil)
    ref = ref[0].permute(1, 2, 0).numpy()
    dx = torch.from_numpy(x).cuda()
    row = []
    for name, kw in (("fp32", {}), ("b3", {"precision": 2}), ("b3 wr2", {"precision": 2, "_tile": 4}), ("b3 m", {"precision": 2, "_tile": 5})):
        tile = kw.pop("_tile", -1)
        out = torch.full((H, W, Cout), 7e7, device="cuda")
        o = lib.opts(**kw)
        lib.check(lib.tdnet_op_conv2d(dx.data_ptr(), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, KS, 1, dil, None, 0, ctypes.byref(o), tile, out.data_ptr(), None))
        e = np.abs(out.cpu().numpy().astype(np.float64) - ref)
        row.append("%s max %.3e rms %.3e" % (name, e.max(), np.sqrt((e ** 2).mean())))
    print("conv %dx%d %d->%d k%d d%d vs fp64: " % (H, W, Cin, Cout, KS, dil) + " | ".join(row), flush=True)


accuracy(64, 128, 256, 256, 3, 2)
accuracy(64, 128, 512, 512, 3, 4)
accuracy(97, 193, 512, 512, 1, 1)
accuracy(33, 65, 128, 132, 3, 1)

SHAPES = [("layer4 512->512 d4", 128, 256, 512, 512, 4), ("layer4 256->512 d4", 128, 256, 256, 512, 4), ("layer3 256->256 d2", 128, 256, 256, 256, 2),
          ("head 512->128", 128, 256, 512, 128, 1), ("layer2 128->128", 128, 256, 128, 128, 1),
          ("native l4 512->512 d4", 97, 193, 512, 512, 4), ("native l3 256->256 d2", 97, 193, 256, 256, 2)]
VAR = [("fp32", {"overlap": 40}, -1), ("fp32 chunks", {"overlap": 45}, -1), ("b3 auto", {"precision": 2, "overlap": 40}, -1), ("b3 wr4", {"precision": 2, "overlap": 40}, 3),
       ("b3 wr2", {"precision": 2, "overlap": 40}, 4), ("b3 m", {"precision": 2, "overlap": 40}, 5), ("b3 chunks wr4", {"precision": 2, "overlap": 45}, 3), ("b3 chunks m", {"precision": 2, "overlap": 45}, 5)]
for (nm, H, W, Cin, Cout, d) in SHAPES:
    row = []
    for (vn, kw, tile) in VAR:
        if (kw["overlap"] & 1) and d % 2:
            continue
        o = lib.opts(**kw)
        ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, 3, 1, d, tile, 20, ctypes.byref(o), None) for _ in range(2))
        row.append("%s %.3f" % (vn, ms))
    print("%-24s " % nm + " | ".join(row), flush=True)
for (nm, H, W, Cin, Cout) in (("enc_v 512->512 1x1", 128, 256, 512, 512), ("fc 512->512 on Lk", 32, 64, 512, 512), ("enc_q0 512->64 1x1", 128, 256, 512, 64)):
    row = []
    for (vn, kw, tile) in (("fp32", {}, -1), ("b3 auto", {"precision": 2}, -1), ("b3 wr4", {"precision": 2}, 3), ("b3 wr2", {"precision": 2}, 4), ("b3 m", {"precision": 2}, 5)):
        o = lib.opts(**kw)
        ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, 1, 1, 1, tile, 20, ctypes.byref(o), None) for _ in range(2))
        row.append("%s %.4f" % (vn, ms))
    print("%-24s " % nm + " | ".join(row), flush=True)

# ---- schedule variants / skip probes of the 256-row Winograd GEMM (a -DTD_B3_PROBE build only: TDNET_EXTRA_CXXFLAGS=-DTD_B3_PROBE) ----
if os.environ.get("TDNET_EXTRA_CXXFLAGS", "").find("TD_B3_PROBE") >= 0:
    o = lib.opts(precision=2, overlap=40)
    for (var, skip) in ((0, 0), (1, 0), (8, 0), (0, 64), (0, 66), (0, 67), (0, 125), (0, 3), (0, 61)):
        os.environ["TD_B3_VAR"] = str(var); os.environ["TD_B3_SKIP"] = str(skip)
        row = []
        for (nm, H, W, Cin, Cout, d) in SHAPES[:3]:
            ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, 3, 1, d, 3, 20, ctypes.byref(o), None) for _ in range(2))
            row.append("%s %.3f" % (nm, ms))
        print("var %d skip %2d: " % (var, skip) + " | ".join(row), flush=True)
