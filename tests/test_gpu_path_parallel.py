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
