"""world_size-2 gloo run of the multi-GPU plumbing (weight broadcast, clip sharding, confusion-matrix all-reduce)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tdnet_amd import arch, parallel, weights


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    spec = arch.model_spec("td2", 19, "resnet18")
    h, wd = 5, 9
    sd = weights.synth_state_dict(spec, h, wd, 0) if rank == 0 else None
    got = parallel.broadcast_state_dict(spec, h, wd, sd, torch.device("cpu"))
    ref = weights.synth_state_dict(spec, h, wd, 0)
    ok = all(np.array_equal(np.asarray(ref[k], np.float32).reshape(-1), np.asarray(got[k], np.float32).reshape(-1)) for k in ref)
    clips = parallel.clips_of_rank(5, rank, world)
    hist = torch.zeros(19, 19, dtype=torch.int64)
    hist[rank, rank] = 10 + rank
    parallel.allreduce_sum(hist)
    t = parallel.allreduce_max(torch.tensor([1.0 + rank]))
    parallel.barrier()
    q.put((rank, ok, clips, int(hist.sum()), float(t)))
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]                                 # identical weights on both ranks
    assert res[0][2] == [0, 2, 4] and res[1][2] == [1, 3]          # every clip exactly once
    assert res[0][3] == res[1][3] == 21                            # summed confusion matrix
    assert res[0][4] == res[1][4] == 2.0                           # max over ranks
