#!/usr/bin/env python3
"""GPU-box: same-process A/B of an ENVIRONMENT switch read at launch time by an experimental kernel path (e.g. TD_B3_NA3=1), interleaved rounds
on one handle.    python tools/env_ab.py VAR [HxW] [opts k=v,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdnet_amd import arch, weights
from tdnet_amd.model import td4_psp18
var = sys.argv[1]
H, W = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1024x2048").split("x"))
opts = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[3].split(","))} if len(sys.argv) > 3 else {"precision": 2}
dev = torch.device("cuda", 0)
spec = arch.model_spec("td4", 19, "resnet18")
sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
clip = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, 8, seed=100)]
m = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None, kernel_opts=opts).eval().to(dev)
m.load_state_dict(sd)
res = {"off": [], "on": []}
outs = {}
with torch.no_grad():
    t = 0
    for _ in range(8):
        m(clip[t % 8], pos_id=t % 4); t += 1
    for rnd in range(4):
        for tag in ("off", "on"):
            if tag == "on": os.environ[var] = "1"
            else: os.environ.pop(var, None)
            for _ in range(8):
                m(clip[t % 8], pos_id=t % 4); t += 1
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(60):
                o = m(clip[t % 8], pos_id=t % 4); t += 1
            torch.cuda.synchronize()
            res[tag].append(60 / (time.perf_counter() - t0))
            outs[tag] = o.clone()
print("%s %dx%d %s  off: %s | on: %s frames/s | last logits equal: %s" % (var, H, W, opts, " ".join("%.1f" % x for x in res["off"]), " ".join("%.1f" % x for x in res["on"]), bool(torch.equal(outs["off"], outs["on"]))))
