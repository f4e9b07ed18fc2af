"""Host-side cost of enqueueing one frame (101 kernel launches + events): perf_counter around model() right after a device sync,
when the queue is empty and the call returns as soon as everything is enqueued."""
import sys, time; sys.path.insert(0, ".")
import torch
from tdnet_amd import arch, weights
from tdnet_amd.model import td4_psp18
dev = torch.device("cuda:0")
H, W = 1024, 2048
spec = arch.model_spec("td4", 19, "resnet18")
m = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None).eval().to(dev)
m.load_state_dict(weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0))
clip = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, 4, seed=1)]
with torch.no_grad():
    for t in range(8): m(clip[t % 4], pos_id=t % 4)
    torch.cuda.synchronize()
    cpu, tot = [], []
    for t in range(8, 28):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m(clip[t % 4], pos_id=t % 4)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        cpu.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print("enqueue (host) ms/frame: min %.3f median %.3f   enqueue+execute ms/frame: min %.3f median %.3f" %
      (min(cpu), sorted(cpu)[len(cpu) // 2], min(tot), sorted(tot)[len(tot) // 2]))
