"""K sweep of the persistent 1x1 GEMM: separates per-tile overhead (falls with K) from per-step throughput (does not)."""
import sys; sys.path.insert(0, ".")
import torch
from tdnet_amd import _capi
lib = _capi.lib(); torch.zeros(1, device="cuda")
for pers in (1, 0):
    lib.tdnet_set_gemm_persistent(pers)
    for (H, W) in ((128, 256), (64, 128)):
        for Cin in (128, 256, 512, 1024, 2048, 4096):
            Cout = 512
            ms = min(lib.tdnet_bench_conv(H, W, Cin, Cout, 1, 1, 1, 3, 20, None) for _ in range(3))
            tf = 2.0 * H * W * Cin * Cout / ms / 1e9
            print("persistent=%d M=%6d K=%4d N=%d: %.4f ms  %.1f TF" % (pers, H * W, Cin, Cout, ms, tf), flush=True)
