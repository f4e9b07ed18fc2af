#!/usr/bin/env python3
"""GPU-box probe: the chunk-aware Winograd transform kernels (td_wino.h k_wino4_*_c<VW>) with 1, 2, 4 channels per lane, whole convs
(overlap bit 2) and row-parity chunks (bit 1), against torch fp32 -- where (pixel row parity / channel / position) are the errors?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from tdnet_amd import _capi
lib = _capi.Lib(sys.argv[1]) if len(sys.argv) > 1 else _capi.test_lib()
only = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else None
print(lib.tdnet_version().decode())
for (H, W, Cin, Cout, dil, resid) in ((13, 21, 64, 128, 2, True), (32, 64, 256, 256, 2, True), (32, 64, 256, 256, 2, False), (16, 32, 128, 128, 1, False)):
    g = np.random.default_rng(0)
    x = g.standard_normal((H, W, Cin)).astype(np.float32)
    w = (g.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = g.standard_normal(Cout).astype(np.float32)
    r = g.standard_normal((H, W, Cout)).astype(np.float32) if resid else None
    ref = F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(w), torch.from_numpy(b), 1, dil, dil)
    if resid:
        ref = ref + torch.from_numpy(r).permute(2, 0, 1)[None]
    ref = F.relu(ref)[0].permute(1, 2, 0).numpy()
    dx = torch.from_numpy(x).cuda(); dr = torch.from_numpy(r).cuda() if resid else None
    for ov in (only or (0, 2, 2 | 16, 2 | 32, 1, 1 | 16, 1 | 32)):
        out = torch.full((H, W, Cout), 7e7, device="cuda")
        o = lib.opts(winograd=4, overlap=ov)
        lib.check(lib.tdnet_op_conv2d(dx.data_ptr(), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, 3, 1, dil, dr.data_ptr() if resid else None, 1,
                                      ctypes.byref(o), -1, out.data_ptr(), None))
        torch.cuda.synchronize()
        e = np.abs(out.cpu().numpy() - ref)
        bad = e > 1e-3
        msg = ""
        if bad.any():
            ys, xs, cs = np.nonzero(bad)
            msg = " BAD %d of %d: rows %s.. parity %s, cols %s.., channels %s.. (mod 4: %s, mod 64 min %d max %d), stale 7e7: %d" % (
                bad.sum(), bad.size, sorted(set(ys.tolist()))[:6], sorted(set((ys % 2).tolist())), sorted(set(xs.tolist()))[:6],
                sorted(set(cs.tolist()))[:8], sorted(set((cs % 4).tolist())), (cs % 64).min(), (cs % 64).max(), int((out.cpu().numpy() > 1e7).sum()))
        print("%dx%d %d->%d d%d resid %d overlap %2d: max err %.3e%s" % (H, W, Cin, Cout, dil, resid, ov, e.max(), msg))
