#!/usr/bin/env python3
"""GPU-box probe: the attention kernels at the frame's shapes -- exact fp32 MFMA (online 2), fp16 MFMA (16), bf16x3 split (17): error against an fp64
evaluation (torch on the GPU, plumbing) and, under rocprofv3 --kernel-trace --stats, the kernels' own durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
for (Lq, Lk, DV, scale) in ((32768, 2048, 512, 0.5), (18721, 1225, 512, 0.5), (32768, 2048, 128, 0.5), (8192, 2048, 512, 2.0)):
    q = (torch.randn(Lq, 64, generator=g) * scale).to(dev); k = (torch.randn(Lk, 64, generator=g) * scale).to(dev)
    vp = torch.zeros((Lk + 127) // 128 * 128, DV); vp[:Lk] = torch.randn(Lk, DV, generator=g); vp = vp.to(dev)
    b = torch.randn(DV, generator=g).to(dev); r = torch.randn(Lq, DV, generator=g).to(dev)
    ref = torch.softmax(q.double() @ k.double().t() / 8.0, dim=1) @ vp[:Lk].double() + b.double() + r.double()
    s = torch.cuda.current_stream().cuda_stream
    row = []
    for online, name in ((2, "fp32"), (16, "fp16"), (17, "bf16x3")):
        out = torch.empty(Lq, DV, device=dev)
        for _ in range(4):
            lib.check(lib.tdnet_op_attention(q.data_ptr(), k.data_ptr(), vp.data_ptr(), b.data_ptr(), r.data_ptr(), Lq, Lk, DV, online | (32 if online < 16 else 0), None, None, None, out.data_ptr(), s))
        e = (out.double() - ref).abs()
        row.append("%s max %.2e rms %.2e" % (name, e.max().item(), e.pow(2).mean().sqrt().item()))
    print("Lq %d Lk %d DV %d |q|,|k| ~ %.1f: " % (Lq, Lk, DV, scale) + " | ".join(row), flush=True)
