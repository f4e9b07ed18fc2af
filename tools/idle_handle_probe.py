#!/usr/bin/env python3
"""GPU-box probe: why do IDLE handles of one process slow a busy one (228 instead of 333 frames/s)?

A handle owns up to two internal HIP streams besides the caller's; HIP (ROCclr) maps a process's streams onto GPU_MAX_HW_QUEUES hardware
queues (default 4) round robin, so with enough streams alive the busy handle's two row-parity chains share ONE hardware queue and serialise.
This times the C2 workload (td2-psp18 @1024x2048) with 0 / 1 / 2 / 3 idle td4 handles alive; run it under different GPU_MAX_HW_QUEUES:
    for q in 4 8 16; do GPU_MAX_HW_QUEUES=$q python tools/idle_handle_probe.py; done"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tdnet_amd import arch, weights  # noqa: E402
from tdnet_amd.model import td2_psp50, td4_psp18  # noqa: E402

H, W = 1024, 2048
dev = torch.device("cuda", 0)
spec2, spec4 = arch.model_spec("td2", 19, "resnet18"), arch.model_spec("td4", 19, "resnet18")
sd2 = weights.synth_state_dict(spec2, arch.feat_size(H), arch.feat_size(W), 0)
sd4 = weights.synth_state_dict(spec4, arch.feat_size(H), arch.feat_size(W), 0)
clip = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, 4, seed=7)]
idle = []
print("GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES", "(default 4)"))
for n_idle in range(4):
    m = td2_psp50.td2_psp50(nclass=19, path_num=2, model_path=None, backbone="resnet18").eval().to(dev)
    m.load_state_dict(sd2)
    with torch.no_grad():
        for t in range(8):
            m(clip[t % 4], pos_id=t % 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(80):
            m(clip[t % 4], pos_id=t % 2)
        torch.cuda.synchronize()
    print("  %d idle td4 handle(s) alive: %.1f frames/s" % (n_idle, 80 / (time.perf_counter() - t0)), flush=True)
    m.engine.close()
    del m
    k = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None).eval().to(dev)   # one more idle handle: built, run once, kept
    k.load_state_dict(sd4)
    with torch.no_grad():
        k(clip[0], pos_id=0)
    torch.cuda.synchronize()
    idle.append(k)
