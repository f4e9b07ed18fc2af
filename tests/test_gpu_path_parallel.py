"""SURVEY 8f-N4 on the GPU: ONE video stream served by two ranks (parallel.PathParallelStream; rank g takes frames t = g mod 2,
cache entries exchanged between tdnet_encode and tdnet_propagate) is held to the CHECKER -- the CPU oracle on the same stream,
with the standard gate (max|dlogit| <= 1e-3, tie-band label flips only, mIoU >= 0.9995) -- and, as a size-independent property on
top, must reproduce bit for bit one handle serving the stream.
The GPU boxes have one MI355X, so both ranks share cuda:0 and the exchange runs over gloo (which stages device tensors through
the host); on a real node the same code path uses RCCL all-gather over xGMI."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tdnet_amd import arch, parallel, weights

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(name, H, W, dev):
    from tdnet_amd.model import td2_psp50, td4_psp18
    spec = arch.model_spec(name, 19, "resnet18")
    cls = td4_psp18.td4_psp18 if name == "td4" else td2_psp50.td2_psp50
    m = cls(nclass=19, path_num=spec.path_num, model_path=None, backbone="resnet18").eval().to(dev)
    m.load_state_dict(weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0))
    return spec, m


def _worker(rank, world, port, q, name, H, W, T):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    parallel.init_distributed("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    spec, m = _make(name, H, W, dev)
    frames = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, T, seed=9)]
    with torch.no_grad():
        outs = parallel.PathParallelStream(m, spec.path_num, device=dev, frame_size=(H, W)).process(frames)
    torch.cuda.synchronize()
    assert m.engine is not None and m.engine.fifo_len() == min(T, spec.fifo), (rank, m.engine.fifo_len())   # every rank saw every entry
    q.put((rank, {t: o.cpu().numpy() for t, o in outs.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,T", [("td4", 9), ("td2", 5), ("td4", 1)])   # T = 1 < world: rank 1 owns no frame, never encodes, still pushes
def test_two_ranks_one_stream_bit_identical(name, T):
    H, W = 129, 257
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, name, H, W, T)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res[0]) == list(range(0, T, 2)) and sorted(res[1]) == list(range(1, T, 2))
    dev = torch.device("cuda", 0)
    spec, m = _make(name, H, W, dev)
    from oracle import tdnet_ref                                   # the checker: the reference's op graph on the CPU
    import test_gpu_model as tm
    oracle = tdnet_ref.TDNetRef(spec, weights.synth_state_dict(spec, arch.feat_size(H), arch.feat_size(W), 0))
    hist = np.zeros((19, 19), np.int64)
    with torch.no_grad():
        for t, x in enumerate(weights.synth_video(H, W, T, seed=9)):
            got = res[t % 2][t]
            exp = oracle.forward(torch.from_numpy(x), t % spec.path_num).numpy()
            tm.check_frame(got, exp, ("path-parallel x2", name, t), hist)       # two ranks vs the oracle
            ref = m(torch.from_numpy(x).to(dev), pos_id=t % spec.path_num).cpu().numpy()
            assert np.array_equal(got, ref), (t, float(np.abs(got - ref).max()))   # and bit-identical to one handle
    assert tm.clip_miou(hist) >= 0.9995


@pytest.mark.parametrize("name,T", [("td4", 11), ("td2", 5)])
def test_two_frames_in_flight_on_one_gpu_bit_identical(name, T):
    """parallel.FramePipelinedStream: the same round with the ranks replaced by two LANES of one process (two handles, two HIP streams,
    the exchange a device-to-device copy) -- one clip, two frames in flight.  Bit for bit one handle serving the stream, over an odd
    number of frames (a short last round), a second process() call that continues the stream without a join in between, and labels."""
    H, W = 129, 257
    dev = torch.device("cuda", 0)
    spec, m = _make(name, H, W, dev)
    _, a = _make(name, H, W, dev)
    _, b = _make(name, H, W, dev)
    P = spec.path_num
    frames = [torch.from_numpy(x).to(dev) for x in weights.synth_video(H, W, T + 4, seed=9)]
    with torch.no_grad():
        ref = [m(x, pos_id=t % P) for t, x in enumerate(frames)]
        fp = parallel.FramePipelinedStream([a, b], P, dev, (H, W))
        outs = fp.process(frames[:T], join=False)                       # T odd: lane 1 has no frame in the last round
        outs += fp.process(frames[T:T + 1], first_frame=T, join=False)  # ... and the next call starts on lane 0 again
        labs = fp.process(frames[T + 1:], labels=True, first_frame=T + 1)
        torch.cuda.synchronize()
    assert a.engine.fifo_len() == b.engine.fifo_len() == spec.fifo
    for t in range(T + 1):
        assert torch.equal(outs[t], ref[t]), (name, t, float((outs[t] - ref[t]).abs().max()))
    for i, lab in enumerate(labs):
        assert torch.equal(lab[0].long(), ref[T + 1 + i][0].max(0)[1]), (name, T + 1 + i)
    # a failing frame (wrong size) resets the stream instead of leaving the two FIFOs in different states; the next clip runs as from new
    with torch.no_grad():
        with pytest.raises(Exception):
            fp.process([frames[0], frames[1][:, :, :65]], first_frame=0)
        assert a.engine.fifo_len() == b.engine.fifo_len() == 0
        again = fp.process(frames[:3])
        torch.cuda.synchronize()
    for t in range(3):
        assert torch.equal(again[t], ref[t]), (name, "after reset", t)
