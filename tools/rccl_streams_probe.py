#!/usr/bin/env python3
"""GPU-box probe: an initialised RCCL communicator in the process (its streams / queues) and the frame rate of a handle.
    rccl_streams_probe.py [mode] [k=v,...]      modes: none | nccl_first (bench.py's order at N > 1) | handle_first (handle built and run BEFORE init)
World size 1 over the nccl backend, one all-reduce and one broadcast to force the communicator."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from tdnet_amd import arch, weights  # noqa: E402
from tdnet_amd.model import td4_psp18  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
opts = {}
if len(sys.argv) > 2:
    for part in sys.argv[2].split(","):
        k, _, v = part.partition("=")
        opts[k] = int(v)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def nccl():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(1 << 20, device=dev)
    dist.all_reduce(t); dist.broadcast(t, src=0)
    torch.cuda.synchronize()


H, W = 1024, 2048
spec = arch.model_spec("td4", 19, "resnet18")
sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
clip = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, 4, seed=7)]


def rate(m, n=80):
    with torch.no_grad():
        for t_ in range(10):
            m(clip[t_ % 4], pos_id=t_ % 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t_ in range(n):
            m(clip[t_ % 4], pos_id=t_ % 4)
        torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


if mode == "nccl_first":
    nccl()
m = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, kernel_opts=opts).eval().to(dev)
m.load_state_dict(sd)
r1 = rate(m)
msg = "%s %s: %.1f frames/s" % (mode, opts or "", r1)
if mode == "handle_first":
    nccl()
    msg += "; after nccl init: %.1f" % rate(m)
    m2 = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, kernel_opts=opts).eval().to(dev)
    m2.load_state_dict(sd)
    m.engine.close()
    msg += "; a second handle created after it: %.1f" % rate(m2)
print(msg)
if dist.is_initialized():
    dist.destroy_process_group()
