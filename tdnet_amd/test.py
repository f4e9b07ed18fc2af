#!/usr/bin/env python3
"""Frame-loop harness with the reference's command line (Testing/test.py:85-107) on the MI355X package.

    python -m tdnet_amd.test --model td4-psp18 --img_path /path/to/frames --output_path ./output/ \\
                             --_td4_psp18_path ./checkpoint/td4-psp18.pkl

Same loop as Testing/test.py:45-81: pos_id = i % path_num, forward timed between two device synchronisations, frames
i > 5 averaged, argmax = output.max(1)[1], quarter-resolution colour PNG per frame.  Differences: PNG I/O through PIL
(imageio / cv2 are not in this image, so no on-screen display); `--gpu` sets HIP_VISIBLE_DEVICES as well;
`--synthetic_seed N` runs on seeded synthetic weights when no checkpoint is available; `--in_size HxW` (default 769x1537,
test.py:24) must match the checkpoint's LayerNorm shape exactly as in the reference.
"""
import argparse
import os
import timeit

import numpy as np
import torch


def test(args):
    os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu                      # test.py:19
    os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)
    from tdnet_amd.dataloader import cityscapesLoader
    from tdnet_amd.model import td2_psp50, td4_psp18
    device = torch.device("cuda")
    H, W = (int(v) for v in args.in_size.lower().split("x"))
    vid_seq = cityscapesLoader(img_path=args.img_path, in_size=(H, W), pin_memory=getattr(args, "prefetch", False))
    vid_seq.load_frames()
    if args.model == "td4-psp18":
        path_num = 4
        model = td4_psp18.td4_psp18(nclass=19, path_num=path_num, model_path=args._td4_psp18_path, synthetic_seed=args.synthetic_seed)
    elif args.model in ("td2-psp50", "td2-psp18", "td2-psp34"):
        path_num = 2
        model = td2_psp50.td2_psp50(nclass=19, path_num=path_num, model_path=args._td2_psp50_path,
                                    backbone="resnet" + args.model[-2:], synthetic_seed=args.synthetic_seed)
    elif args.model == "psp101":                                        # test.py:34-38
        path_num = 1
        from tdnet_amd.model import pspnet
        model = pspnet.pspnet(nclass=19, model_path=args._psp101_path, synthetic_seed=args.synthetic_seed)
    else:
        raise SystemExit("model must be one of td4-psp18, td2-psp50, td2-psp18, td2-psp34, psp101")
    model.eval()
    model.to(device)

    def save(pred, img_name, folder, ori_size):
        pred = np.squeeze(pred, axis=0).astype(np.int8)
        # cv2.resize(pred, (W//4, H//4), INTER_NEAREST) (test.py:64): nearest sample at floor(dst * scale)
        oh, ow = ori_size[1] // 4, ori_size[0] // 4
        ys = np.minimum((np.arange(oh) * (pred.shape[0] / oh)).astype(np.int64), pred.shape[0] - 1)
        xs = np.minimum((np.arange(ow) * (pred.shape[1] / ow)).astype(np.int64), pred.shape[1] - 1)
        decoded = vid_seq.decode_segmap(pred[ys][:, xs])
        save_dir = os.path.join(args.output_path, folder)
        os.makedirs(save_dir, exist_ok=True)
        from PIL import Image
        Image.fromarray(decoded.astype(np.uint8)).save(os.path.join(save_dir, img_name))

    timer, i = 0.0, -1
    with torch.no_grad():
        if args.prefetch:
            # throughput loop (not in the reference): frame i + 1 is uploaded while frame i computes, the labels come back
            # asynchronously as int32 (the full-resolution logits are never written), PNGs are written as they arrive
            from tdnet_amd.dataloader import DevicePrefetcher, LabelDownloader
            down = LabelDownloader(device)
            torch.cuda.synchronize()
            start_time = timeit.default_timer()
            for i, (image, img_name, folder, ori_size) in enumerate(DevicePrefetcher(vid_seq.data, device)):
                labels = model.forward_labels(image, pos_id=i % path_num)
                for tag, pred in down.submit(labels, (img_name, folder, ori_size)):
                    save(pred, *tag)
            for tag, pred in down.drain():
                save(pred, *tag)
            torch.cuda.synchronize()
            timer = timeit.default_timer() - start_time
            print("---------------------")
            print(" Model: {0:s}".format(args.model))
            if i >= 0:
                print(" {0:d} frames, prefetched upload + asynchronous labels: {1:3.5f} s per frame including the PNG writer".format(i + 1, timer / (i + 1)))
            print("---------------------")
            return
        for i, (image, img_name, folder, ori_size) in enumerate(vid_seq.data):
            image = image.to(device)
            torch.cuda.synchronize()
            start_time = timeit.default_timer()
            output = model(image, pos_id=i % path_num)
            torch.cuda.synchronize()
            elapsed_time = timeit.default_timer() - start_time
            if i > 5:
                timer += elapsed_time
            save(output.data.max(1)[1].cpu().numpy(), img_name, folder, ori_size)
            print(" Frame {0:2d}   RunningTime/Latency={1:3.5f} s".format(i + 1, elapsed_time))
    print("---------------------")
    print(" Model: {0:s}".format(args.model))
    if i > 5:
        print(" Average  RunningTime/Latency={0:3.5f} s".format(timer / (i - 5)))
    print("---------------------")


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Params")
    parser.add_argument("--img_path", nargs="?", type=str, default="./data/vid1", help="Path_to_Frame")
    parser.add_argument("--output_path", nargs="?", type=str, default="./output/", help="Path_to_Save")
    parser.add_argument("--_td4_psp18_path", nargs="?", type=str, default="./checkpoint/td4-psp18.pkl", help="Path_to_PSP_Model")
    parser.add_argument("--_td2_psp50_path", nargs="?", type=str, default="./checkpoint/td2-psp50.pkl", help="Path_to_PSP_Model")
    parser.add_argument("--_psp101_path", nargs="?", type=str, default="./checkpoint/psp101.pkl", help="Path_to_PSP_Model")
    parser.add_argument("--gpu", nargs="?", type=str, default="0", help="gpu_id")
    parser.add_argument("--model", nargs="?", type=str, default="td4-psp18", help="model in [td4-psp18, td2-psp50, td2-psp18, td2-psp34]")
    parser.add_argument("--in_size", nargs="?", type=str, default="769x1537", help="HxW fed to the network (test.py:24)")
    parser.add_argument("--synthetic_seed", nargs="?", type=int, default=None, help="run on seeded synthetic weights")
    parser.add_argument("--prefetch", action="store_true", help="throughput loop: upload of the next frame under the current one, asynchronous label download")
    test(parser.parse_args())
