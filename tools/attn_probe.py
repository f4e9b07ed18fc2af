"""Time the attention-propagation kernel alone at the frame's shapes (device-resident inputs, HIP events on the launch stream)."""
import sys; sys.path.insert(0, ".")
import torch
from tdnet_amd import _capi
lib = _capi.test_lib(); dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
def run(Lq, Lk, DV, online=0, iters=20):
    q = (torch.randn(Lq, 64, generator=g) * 0.5).to(dev); k = (torch.randn(Lk, 64, generator=g) * 0.5).to(dev)
    vp = torch.zeros((Lk + 127) // 128 * 128, DV); vp[:Lk] = torch.randn(Lk, DV, generator=g); vp = vp.to(dev)   # padded as the kernels expect
    b = torch.randn(DV, generator=g).to(dev); r = torch.randn(Lq, DV, generator=g).to(dev)
    out = torch.empty(Lq, DV, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.check(lib.tdnet_op_attention(q.data_ptr(), k.data_ptr(), vp.data_ptr(), b.data_ptr(), r.data_ptr(), Lq, Lk, DV, online | 32, None, None, None, out.data_ptr(), s))
    for _ in range(3): call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    fl = 2.0 * Lq * Lk * (64 + DV)
    print("Lq=%6d Lk=%5d DV=%3d %s: %.4f ms  %.1f TF algorithmic" % (Lq, Lk, DV, {0: "two-pass", 1: "online  ", 2: "online-1b"}[online], best, fl / best / 1e9), flush=True)
for shape in ((32768, 2048, 512), (2048, 2048, 512), (18721, 1225, 512), (32768, 2048, 128), (12288, 768, 512)):
    for online in (0, 1, 2):
        run(*shape, online=online)
