"""world_size-2 gloo run of the multi-GPU plumbing (weight broadcast, clip sharding, confusion-matrix all-reduce)."""
import os
import socket

import numpy as np
import pytest
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


# ---- path-parallel single stream (parallel.PathParallelStream): the exchange schedule, on a CPU stand-in for the engine ----
class _ToyStage:
    """Same FIFO semantics as the handle (td4_psp18.py:123-154): entries are pushed in frame order, the output of a frame is a
    function of the FIFO contents at propagate time and of the frame itself, and propagate commits the frame's own entry."""

    def __init__(self, fifo_depth, n=7):
        self.depth, self.n, self.fifo, self.pending = fifo_depth, n, [], None

    def encode(self, img, pos_id=0):
        assert self.pending is None
        f = float(img.sum())
        self.pending = (torch.full((self.n,), f), torch.full((self.n,), f + 0.25 * pos_id), torch.full((2 * self.n,), -f))
        self.cur = (f, pos_id)

    def cache_entry_numel_for(self, H, W):
        return self.n, self.n, 2 * self.n

    def cache_export(self, q, k, v):
        for dst, src in zip((q, k, v), self.pending):
            dst.copy_(src)

    def _commit(self, e):
        self.fifo.append(e)
        if len(self.fifo) > self.depth:
            self.fifo.pop(0)

    def cache_push(self, q, k, v):
        self._commit((q.clone(), k.clone(), v.clone()))

    def propagate(self, labels=False):
        acc = self.cur[0] * 1000.0 + self.cur[1]
        if len(self.fifo) >= self.depth:                               # warm-up frames ignore the cache
            for i, (q, k, v) in enumerate(self.fifo):
                acc += (i + 1) * float(q[0]) + 10.0 * (i + 1) * float(k[0]) + 100.0 * (i + 1) * float(v[0])
        self._commit(self.pending)
        self.pending = None
        return torch.tensor([acc], dtype=torch.float64)


def _toy_frames(T):
    return [torch.full((1, 3, 2, 2), float(t + 1)) for t in range(T)]


def _pp_worker(rank, world, port, q, T, P, depth):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    pp = parallel.PathParallelStream(_ToyStage(depth), P, device=torch.device("cpu"), frame_size=(2, 2))
    outs = pp.process(_toy_frames(T))
    q.put((rank, {t: float(o) for t, o in outs.items()}))
    dist.destroy_process_group()


def _pp_sequential(T, P, depth):
    st, out = _ToyStage(depth), {}
    for t, f in enumerate(_toy_frames(T)):
        st.encode(f, t % P)
        out[t] = float(st.propagate())
    return out


def test_path_parallel_schedule_matches_sequential():
    ctx = mp.get_context("spawn")
    # td4 (FIFO 3) on 2 ranks with a ragged last round; td2 (FIFO 1) on 2 ranks; td4 on 4 ranks = one sub-network per rank,
    # every entry a frame needs comes from a peer (world_size > FIFO depth + 1 is covered by the in-order pushes)
    # (4, 3, ...): a stream SHORTER than one round -- rank 3 owns no frame at all and must still take part in every collective
    # (the geometry comes from frame_size, not from a live engine) instead of failing alone while its peers wait
    for (W, T, P, depth) in ((2, 9, 4, 3), (2, 6, 2, 1), (4, 10, 4, 3), (4, 3, 4, 3)):
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_pp_worker, args=(r, W, port, q, T, P, depth)) for r in range(W)]
        for p in ps:
            p.start()
        res = dict(q.get(timeout=300) for _ in ps)
        for p in ps:
            p.join(timeout=60)
            assert p.exitcode == 0
        with pytest.raises(ValueError):                                # required on every rank, checked before any collective
            parallel.PathParallelStream(_ToyStage(depth), P, rank=0, world=W, device=torch.device("cpu"))
        merged = {}
        for r in range(W):
            assert sorted(res[r]) == list(range(r, T, W))              # rank g serves t = g mod W
            merged.update(res[r])
        assert merged == _pp_sequential(T, P, depth)                   # every rank's FIFO went through the sequential states


# ---- world_size 8: the size the driver's scaling run uses first ------------------------------------
def test_eight_rank_gloo_collectives_and_path_parallel():
    """The same plumbing at world_size 8: weight broadcast into 8 processes, clip sharding, the confusion-matrix / timing reductions,
    gather_strings, and the path-parallel schedule with world 8 > FIFO depth + 1 (every entry a frame needs comes from peers that are
    more than a FIFO away in rank order) incl. a stream shorter than one round and a ragged last round."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert sorted(c for r in res for c in r[2]) == [0, 1, 2, 3, 4]                  # 5 clips on 8 ranks: every clip exactly once, 3 ranks idle
    assert all(r[3] == sum(10 + k for k in range(8)) for r in res)
    assert all(r[4] == 8.0 for r in res)
    for (W, T, P, depth) in ((8, 19, 4, 3), (8, 5, 4, 3), (8, 16, 2, 1)):
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_pp_worker, args=(r, W, port, q, T, P, depth)) for r in range(W)]
        for p in ps:
            p.start()
        out = dict(q.get(timeout=600) for _ in ps)
        for p in ps:
            p.join(timeout=120)
            assert p.exitcode == 0
        merged = {}
        for r in range(W):
            assert sorted(out[r]) == list(range(r, T, W))
            merged.update(out[r])
        assert merged == _pp_sequential(T, P, depth)


def test_init_distributed_has_no_default_port(monkeypatch):
    """A fixed fallback port (29500 until round 3) lets two jobs on one node meet in each other's rendezvous: without MASTER_PORT the
    call must fail loudly, before any process group exists."""
    for k in ("MASTER_PORT", "MASTER_ADDR"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        parallel.init_distributed("gloo")
    assert not dist.is_initialized()


def test_affinity_plan():
    """plan_affinity: ranks split the CPUs of their GPU's NUMA node in rank order, disjointly; unknown nodes share what is allowed."""
    node_cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    got = [parallel.plan_affinity(r, 8, nodes, range(128), node_cpus) for r in range(8)]
    assert [g[0] for g in got] == nodes
    assert all(len(g[1]) == 16 for g in got)
    assert sorted(c for g in got for c in g[1]) == list(range(128))                # disjoint and complete
    assert set(got[5][1]) <= set(node_cpus[1])
    # a cgroup that allows only part of node 0, nothing of node 1: ranks on node 1 fall back to sharing the allowed set
    got = [parallel.plan_affinity(r, 8, nodes, range(8), node_cpus) for r in range(8)]
    assert all(len(g[1]) >= 1 and set(g[1]) <= set(range(8)) for g in got)
    assert sorted(c for g in got[:4] for c in g[1]) == list(range(8))
    assert [g[0] for g in got[4:]] == [-1] * 4
    # no NUMA information at all, fewer CPUs than ranks: everyone still gets a CPU
    got = [parallel.plan_affinity(r, 8, [-1] * 8, range(4), {}) for r in range(8)]
    assert all(len(g[1]) == 1 for g in got)
    assert parallel._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
