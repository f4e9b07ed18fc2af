"""parallel.FramePipelinedStream (and dataloader.DevicePrefetcher) without a GPU: their event / stream logic under a SCHEDULE SIMULATOR.

The class only enqueues work: stage calls, event records, stream waits.  Here torch.cuda's streams and events are replaced by queues of
deferred operations, the stages by a model of a handle (pending entry, FIFO of the last `depth` entries, output = function of the frame
and of the FIFO it met), and the queued operations are then EXECUTED in many legal interleavings -- lane 0 as far ahead as its waits
allow, lane 1 as far ahead, alternating, random.  Whatever the order, every output must be the sequential loop's (test.py:45-53) and
both FIFOs must end in its state; an event wait that is missing (an entry pushed before its frame was encoded, an exchange row
re-exported while a lane still reads it) shows up as a wrong output in the schedules that run one lane ahead."""
import contextlib
import random

import pytest
import torch

from tdnet_amd import parallel


class _Event:
    def __init__(self, sim):
        self.sim, self.done, self.recorded = sim, False, False

    def record(self, stream=None):
        stream = stream or self.sim.current
        assert not self.recorded, "the class under test records an event once"
        self.recorded = True
        stream.ops.append(("record", self))

    def synchronize(self):
        self.sim.run_until(lambda: self.done)

    def query(self):
        return self.done


class _Stream:
    def __init__(self, sim, name):
        self.sim, self.name, self.ops = sim, name, []
        self.cuda_stream = id(self)

    def wait_event(self, ev):
        assert ev.recorded, "waiting for an event that was never recorded would not wait at all"
        self.ops.append(("wait", ev))

    def wait_stream(self, other):
        ev = _Event(self.sim)
        ev.record(other)
        self.wait_event(ev)

    def synchronize(self):
        self.sim.run_until(lambda: not self.ops)


class _Sim:
    """Deferred execution of what the class enqueues; `policy` picks the next stream among those whose head operation can run."""

    def __init__(self, policy, seed=0):
        self.main = _Stream(self, "lane0")
        self.current = self.main
        self.streams = [self.main]
        self.policy, self.rng = policy, random.Random(seed)

    def new_stream(self):
        s = _Stream(self, "lane%d" % len(self.streams))
        self.streams.append(s)
        return s

    def enqueue(self, fn):
        self.current.ops.append(("run", fn))

    @contextlib.contextmanager
    def stream_ctx(self, s):
        prev, self.current = self.current, s
        try:
            yield
        finally:
            self.current = prev

    def _runnable(self):
        return [s for s in self.streams if s.ops and (s.ops[0][0] != "wait" or s.ops[0][1].done)]

    def step(self):
        ready = self._runnable()
        if not ready:
            assert not any(s.ops for s in self.streams), "deadlock: " + str([(s.name, s.ops[0][0]) for s in self.streams if s.ops])
            return False
        if self.policy == "random":
            s = self.rng.choice(ready)
        elif self.policy == "alternate":
            s = ready[self.rng.randrange(len(ready))] if self.rng.random() < 0.2 else ready[0]
            ready.reverse()
        else:                                                         # "first:k" -- lane k as far ahead as its waits allow
            k = int(self.policy.split(":")[1])
            pref = [x for x in ready if x is self.streams[min(k, len(self.streams) - 1)]]
            s = pref[0] if pref else ready[0]
        kind, arg = s.ops.pop(0)
        if kind == "run":
            arg()
        elif kind == "record":
            arg.done = True
        return True

    def run_until(self, cond):
        while not cond():
            assert self.step(), "nothing left to run"

    def drain(self):
        while self.step():
            pass


class _Out:
    def __init__(self):
        self.value = None

    def record_stream(self, s):
        pass


class _Stage:
    """A handle as the split frame sees it: encode leaves an entry pending, propagate reads the FIFO and commits it, push appends a peer's."""

    class _Eng:
        lib = None

    def __init__(self, sim, depth):
        self.sim, self.depth, self.fifo, self.pending, self.engine = sim, depth, [], None, _Stage._Eng()

    def cache_entry_numel_for(self, H, W):
        return 2, 2, 3

    def ensure_engine(self, H, W, device):
        pass

    def encode(self, frame, pos_id=0):
        fid = int(frame)

        def run():
            assert self.pending is None, "a frame was encoded over a pending one"
            self.pending = 1000.0 + fid                              # the entry's content: a function of the frame alone
        self.sim.enqueue(run)

    def cache_export(self, q, k, v):
        def run():
            q.fill_(self.pending); k.fill_(self.pending + 0.25); v.fill_(self.pending + 0.5)
        self.sim.enqueue(run)

    def cache_push(self, q, k, v):
        def run():
            e = float(q[0])
            assert float(k[0]) == e + 0.25 and float(v[-1]) == e + 0.5, "an exchange row was read while it was being rewritten"
            self.fifo = (self.fifo + [e])[-self.depth:]
        self.sim.enqueue(run)

    def propagate(self, labels=False):
        out = _Out()

        def run():
            out.value = (self.pending, tuple(self.fifo))
            self.fifo = (self.fifo + [self.pending])[-self.depth:]
            self.pending = None
        self.sim.enqueue(run)
        return out

    def reset(self):
        self.fifo, self.pending = [], None


def _sequential(T, depth, first=0):
    fifo, outs = [], []
    for t in range(first, first + T):
        e = 1000.0 + t
        outs.append((e, tuple(fifo)))
        fifo = (fifo + [e])[-depth:]
    return outs, fifo


@pytest.fixture
def sim_env(monkeypatch):
    def make(policy, seed):
        sim = _Sim(policy, seed)
        monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: sim.current)
        monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _Event(sim))
        monkeypatch.setattr(torch.cuda, "stream", sim.stream_ctx)
        from tdnet_amd.model import _base
        monkeypatch.setattr(_base._TDNetBase, "_stream_beside", staticmethod(lambda taken, device, lib: sim.new_stream()))
        return sim
    return make


@pytest.mark.parametrize("policy", ["first:0", "first:1", "alternate", "random"])
@pytest.mark.parametrize("depth,P", [(1, 2), (3, 4)])
def test_every_interleaving_gives_the_sequential_loop(sim_env, policy, depth, P):
    for seed in range(6 if policy in ("random", "alternate") else 1):
        sim = sim_env(policy, seed)
        stages = [_Stage(sim, depth), _Stage(sim, depth)]
        fp = parallel.FramePipelinedStream(stages, P, "cpu", (33, 65))
        # three calls without a join in between: an odd count (short last round), a single frame, an even run; then a joined call
        plan, t, outs = [7, 1, 10, 5], 0, []
        for i, n in enumerate(plan):
            outs += fp.process(list(range(t, t + n)), first_frame=t, join=(i == len(plan) - 1))
            t += n
            if seed % 2 == 0 and i == 1:
                sim.drain()                                           # sometimes the device catches up between two calls
        sim.drain()
        want, fifo = _sequential(t, depth)
        assert [o.value for o in outs] == want, (policy, seed)
        assert stages[0].fifo == stages[1].fifo == fifo and stages[0].pending is None and stages[1].pending is None


def test_the_simulator_catches_a_missing_wait(sim_env, monkeypatch):
    """The checker must be able to fail: without the wait for the peer's encode event, a schedule that runs lane 1 ahead pushes an entry
    that has not been exported yet."""
    sim = sim_env("first:1", 0)
    stages = [_Stage(sim, 1), _Stage(sim, 1)]
    fp = parallel.FramePipelinedStream(stages, 2, "cpu", (33, 65))
    real_wait = _Stream.wait_event
    skipped = {"n": 0}

    def leaky(self, ev):
        if self is not sim.main and skipped["n"] < 50:               # lane 1 stops waiting for anything
            skipped["n"] += 1
            return
        real_wait(self, ev)
    monkeypatch.setattr(_Stream, "wait_event", leaky)
    outs = fp.process(list(range(6)), join=False)
    failed = False
    try:
        sim.drain()
        failed = [o.value for o in outs] != _sequential(6, 1)[0]
    except AssertionError:
        failed = True
    assert failed and skipped["n"] > 0


class _DevBuf:
    """A device buffer of DevicePrefetcher: copy_ is an asynchronous upload on the stream that is current when it is issued."""

    def __init__(self, sim):
        self.sim, self.value = sim, None

    def copy_(self, img, non_blocking=False):
        v = float(img.flatten()[0])
        self.sim.enqueue(lambda: setattr(self, "value", v))
        return self


@pytest.mark.parametrize("policy", ["first:0", "first:1", "alternate", "random"])
@pytest.mark.parametrize("depth", [2, 3, 4])
def test_prefetcher_never_overwrites_a_frame_in_use(sim_env, monkeypatch, policy, depth):
    """dataloader.DevicePrefetcher: the consumer's work on frame i (enqueued on its own stream when the frame is yielded) must see frame
    i whatever the interleaving of the copy stream and the consumer's stream -- the copy stream far ahead (uploads want to overwrite
    buffers still unread) and far behind (the consumer wants frames that have not landed)."""
    from tdnet_amd import dataloader
    for seed in range(4 if policy in ("random", "alternate") else 1):
        sim = sim_env(policy, seed)
        monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: sim.new_stream())
        monkeypatch.setattr(dataloader.DevicePrefetcher, "_device_buffer", lambda self, shape: _DevBuf(sim))
        items = [[torch.full((1, 3, 2, 2), float(i)), "f%d" % i, "vid", (2, 2)] for i in range(11)]
        seen = []
        for i, (buf, name, folder, size) in enumerate(dataloader.DevicePrefetcher(items, "cuda", depth=depth)):
            assert name == "f%d" % i
            sim.enqueue(lambda b=buf: seen.append(b.value))          # the consumer's frame: reads the buffer when it EXECUTES
            if seed % 2 and i % 3 == 0:
                sim.drain()
        sim.drain()
        assert seen == [float(i) for i in range(11)], (policy, depth, seed, seen)


@pytest.mark.parametrize("policy", ["first:0", "first:1", "alternate", "random"])
@pytest.mark.parametrize("N", [1, 2, 3, 4])
def test_batch_lanes_wait_for_the_input_and_join_the_caller(sim_env, policy, N):
    """model/_base.py _for_each_sample: the odd samples of a batch run on a second stream.  Whatever the interleaving, a sample must not
    start before the caller's stream has produced the input, and what the caller enqueues AFTER forward() must find every sample done."""
    from tdnet_amd.model._base import _TDNetBase
    for seed in range(3):
        sim = sim_env(policy, seed)
        m = object.__new__(_TDNetBase)
        m._extra_streams, m._extra_streams_for = [], None

        class Eng:
            lib = None
        engines = [Eng() for _ in range(N)]
        m._engines_for_batch = lambda img: engines
        by_raw = lambda raw: next(s for s in sim.streams if s.cuda_stream == raw)
        checked = []
        for rep in range(3):                                          # the second stream is placed once and reused
            st = {"input": False, "done": set()}                      # this batch's own state
            sim.enqueue(lambda st=st: st.__setitem__("input", True))  # the caller's stream produces the batch

            def call(i, eng, raw, st=st):
                def run():
                    assert st["input"], "sample %d started before the input was ready" % i
                    st["done"].add(i)
                by_raw(raw).ops.append(("run", run))
            m._for_each_sample(torch.zeros(N, 3, 2, 2), call)

            def after(st=st):
                assert st["done"] == set(range(N)), "the caller's stream ran ahead of samples %s" % (set(range(N)) - st["done"])
                checked.append(rep)
            sim.enqueue(after)
            if seed == 1:
                sim.drain()
        sim.drain()
        assert len(checked) == 3 and len(sim.streams) == (1 if N == 1 else 2)


class _Labels:
    """A device label map as LabelDownloader sees it."""
    shape, dtype = (1, 2, 2), torch.int32

    def __init__(self, sim, value):
        self.sim, self.value = sim, None
        sim.enqueue(lambda: setattr(self, "value", value))           # the frame's kernel writes it when the caller's stream gets there

    def record_stream(self, s):
        pass


class _HostBuf:
    shape, dtype = (1, 2, 2), torch.int32

    def __init__(self, sim):
        self.sim, self.value = sim, None

    def copy_(self, labels, non_blocking=False):
        self.sim.enqueue(lambda: setattr(self, "value", labels.value))
        return self

    def numpy(self):
        return self.value


@pytest.mark.parametrize("policy", ["first:0", "first:1", "alternate", "random"])
def test_label_downloader_returns_every_map_once_and_in_order(sim_env, monkeypatch, policy):
    """dataloader.LabelDownloader: the download of frame t must wait for frame t's kernel, results come back in order, exactly once, and
    a pinned buffer is not reused for a new download while its previous content has not been handed to the caller."""
    from tdnet_amd import dataloader
    for seed in range(3):
        sim = sim_env(policy, seed)
        monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: sim.new_stream())
        monkeypatch.setattr(dataloader.LabelDownloader, "_host_buffer", staticmethod(lambda shape, dtype: _HostBuf(sim)))
        down = dataloader.LabelDownloader("cuda", depth=3)
        got = []
        for t in range(13):
            for tag, arr in down.submit(_Labels(sim, 100 + t), t):
                got.append((tag, arr))                                # consumed at once, as the contract asks
            if seed == 1 and t % 4 == 0:
                sim.drain()
        got += [(tag, arr) for tag, arr in down.drain()]
        assert got == [(t, 100 + t) for t in range(13)], (policy, seed, got)
