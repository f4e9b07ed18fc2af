#!/usr/bin/env python3
"""GPU-box probe: the loader-side classes on synthetic frames at full size -- frames/s of
  literal     for item in data: model(item.to(device)).max(1)[1].cpu()                       (test.py:47-61)
  prefetched  for item in DevicePrefetcher(data): LabelDownloader.submit(model.forward_labels(item))   (tdnet_amd/dataloader.py)
and that both produce the same labels.   python tools/pcie_loader_probe.py [--size 1024x2048] [--frames 48]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1024x2048")
    ap.add_argument("--frames", type=int, default=48)
    ap.add_argument("--pinned", action="store_true", help="the frames are page-locked already (no bounce copy)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from tdnet_amd import arch, weights
    from tdnet_amd.dataloader import DevicePrefetcher, LabelDownloader
    from tdnet_amd.model import td4_psp18
    H, W = (int(v) for v in a.size.lower().split("x"))
    spec = arch.model_spec("td4", 19, "resnet18")
    sd = weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0)
    dev = torch.device("cuda", 0)
    base = [torch.from_numpy(x) for x in weights.synth_video(H, W, 8, seed=100)]
    if a.pinned:
        base = [x.pin_memory() for x in base]                         # what cityscapesLoader(..., pin_memory=True) hands out
    data = [[base[t % 8], "f%04d.png" % t, "vid", (W, H)] for t in range(a.frames)]
    m = td4_psp18.td4_psp18(nclass=19, path_num=4, model_path=None).eval().to(dev)
    m.load_state_dict(sd)
    with torch.no_grad():
        for t in range(8):
            m(data[t][0].to(dev), pos_id=t % 4)
        m.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lit = [m(img.to(dev), pos_id=t % 4).max(1)[1].cpu().numpy() for t, (img, *_r) in enumerate(data)]
        torch.cuda.synchronize()
        fps_lit = a.frames / (time.perf_counter() - t0)
        m.reset()
        down = LabelDownloader(dev)
        got = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t, (img, name, folder, size) in enumerate(DevicePrefetcher(data, dev)):
            for tag, arr in down.submit(m.forward_labels(img, pos_id=t % 4), t):
                got[tag] = int(arr.astype(np.int64).sum())            # consumed at once (the array is a view of a pinned buffer)
        for tag, arr in down.drain():
            got[tag] = int(arr.astype(np.int64).sum())
        torch.cuda.synchronize()
        fps_pre = a.frames / (time.perf_counter() - t0)
    same = all(got[t] == int(lit[t].astype(np.int64).sum()) for t in range(a.frames))
    print("td4-psp18 %dx%d fp32, %d frames from host memory: literal loop %.1f frames/s, DevicePrefetcher + LabelDownloader %.1f frames/s (GPU_MAX_HW_QUEUES=%s, frames %s); label sums %s" %
          (H, W, a.frames, fps_lit, fps_pre, os.environ.get("GPU_MAX_HW_QUEUES"), "page-locked" if a.pinned else "pageable", "identical" if same else "DIFFER"))


if __name__ == "__main__":
    main()
