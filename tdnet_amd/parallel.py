"""Multi-GPU: independent video clips sharded over ranks, one process per GPU.

A clip's only state is its own K/Q/V FIFO (td4_psp18.py:118-134), so ranks never exchange per-frame data.  The only
collectives are: ONE broadcast of the flat fp32 weight blob from rank 0 (RCCL over xGMI; 219 MB for td4) at start, and
small all-reduces of the 19x19 confusion matrix / timing at the end.  Backend "nccl" is RCCL on ROCm; the CPU tests run
the same code over gloo with world_size 2.

PathParallelStream is the other mode of SURVEY.md §8e (row N4): ONE video stream served by W ranks, rank g taking the frames
t = g (mod W).  That path has a real exchange step -- every frame's (q,k,v) cache entry, 5.2 MB at 1024x2048 -- and it is the
only per-frame collective in the package: one all-gather of W entries per round of W frames.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import arch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    """One process per GPU; rendezvous from the launcher's environment (torch.distributed.run / bench.py's self-launch).  There is
    no default port: two jobs on one node falling back to the same fixed port would join each other's rendezvous or fail to bind."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            raise RuntimeError("tdnet_amd.parallel.init_distributed: WORLD_SIZE=%d but MASTER_PORT is not set -- start the ranks with "
                               "`python -m torch.distributed.run --master-addr 127.0.0.1 --master-port <free port> ...` or "
                               "`python bench.py --gpus N` (which picks a free port itself)" % world)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node of the GPU behind HIP ordinal `device_index` (PCI address from the device properties ->
    /sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node), or -1 when the platform does not say."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")) as f:
            return int(f.read().strip())
    except Exception:
        return -1


def plan_affinity(local_rank, local_world, node_of_rank, allowed, node_cpus):
    """Pure planning (tested on the CPU): the CPUs rank `local_rank` should run on.  Ranks whose GPUs share a NUMA node split that
    node's allowed CPUs evenly, in rank order; a rank whose node is unknown (-1) or has no allowed CPU shares ALL allowed CPUs
    evenly with the other such ranks.  Every rank gets at least one CPU; the sets of two ranks are disjoint whenever there are
    at least as many CPUs as ranks in the group."""
    allowed = sorted(allowed)
    node = node_of_rank[local_rank]
    pool = [c for c in node_cpus.get(node, []) if c in set(allowed)] if node >= 0 else []
    if pool:
        group = [r for r in range(local_world) if node_of_rank[r] == node]
    else:
        node = -1
        group = [r for r in range(local_world) if node_of_rank[r] < 0 or not [c for c in node_cpus.get(node_of_rank[r], []) if c in set(allowed)]]
        pool = allowed
    i, n = group.index(local_rank), len(group)
    lo, hi = i * len(pool) // n, (i + 1) * len(pool) // n
    mine = pool[lo:hi] or [pool[min(lo, len(pool) - 1)]]
    return node, mine


def pin_rank(local_rank, local_world, device_indices=None, sysfs="/sys", apply=True):
    """Pins this process to the cores next to its GPU (8 launcher processes, each issuing ~25 k kernel launches per
    second, must not migrate across sockets or sit on each other's cores) and sets the intra-op thread count to match
    (torch.distributed.run exports OMP_NUM_THREADS=1 to every rank).  Returns what was done, for the bench line."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:                                             # not Linux
        return {"pinned": False, "why": "sched_getaffinity unavailable"}
    dev = list(range(local_world)) if device_indices is None else list(device_indices)
    nodes = [gpu_numa_node(d, sysfs) if torch.cuda.is_available() else -1 for d in dev]
    node_cpus = {}
    for nd in set(nodes):
        if nd >= 0:
            try:
                with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % nd)) as f:
                    node_cpus[nd] = _parse_cpulist(f.read())
            except OSError:
                node_cpus[nd] = []
    node, cpus = plan_affinity(local_rank, local_world, nodes, allowed, node_cpus)
    info = {"pinned": False, "numa_node": node, "cpus": "%d-%d" % (cpus[0], cpus[-1]) if cpus == list(range(cpus[0], cpus[-1] + 1)) else ",".join(map(str, cpus)),
            "n_cpus": len(cpus)}
    if apply:
        try:
            os.sched_setaffinity(0, cpus)
            info["pinned"] = True
        except OSError as e:
            info["why"] = str(e)
    nthr = max(1, min(len(cpus), 32))
    os.environ["OMP_NUM_THREADS"] = str(nthr)
    try:
        torch.set_num_threads(nthr)
    except RuntimeError:
        pass
    info["omp_num_threads"] = nthr
    return info


def gather_strings(text, world, device):
    """Every rank contributes a short ASCII string; every rank returns the list (fixed 256-byte rows through one all-reduce)."""
    row = torch.zeros(world, 256, dtype=torch.int64, device=device)
    b = text.encode("ascii", "replace")[:256]
    rank = dist.get_rank() if dist.is_initialized() else 0
    row[rank, :len(b)] = torch.tensor(list(b), dtype=torch.int64)
    row = allreduce_sum(row).cpu()
    return [bytes(int(v) for v in r if v).decode("ascii") for r in row]


def clips_of_rank(n_clips, rank, world):
    """clip c is served by rank c mod world; frames of a clip stay ordered on one GPU."""
    return list(range(rank, n_clips, world))


def broadcast_state_dict(spec, h, w, sd, device, src=0):
    """Rank `src` holds `sd` ({name: ndarray}); every rank returns the identical dict.  One flat fp32 broadcast."""
    shapes = arch.state_dict_shapes(spec, h, w)
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return sd
    sizes = {k: int(np.prod(s)) if len(s) else 1 for k, s in shapes.items()}
    total = sum(sizes.values())
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        off = 0
        host = np.empty(total, np.float32)
        for k in shapes:
            host[off:off + sizes[k]] = np.asarray(sd[k], dtype=np.float32).reshape(-1)
            off += sizes[k]
        flat.copy_(torch.from_numpy(host))
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out, off = {}, 0
    for k, s in shapes.items():
        out[k] = host[off:off + sizes[k]].reshape(s).copy()
        off += sizes[k]
    return out


def allreduce_sum(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class PathParallelStream:
    """One video stream on W ranks (SURVEY.md §8e "alternative", §8f N4).

    Frame t is served by rank t mod W with sub-network t mod P.  What frame t needs from its predecessors is only their cache
    entries (td4_psp18.py:145-147 read K/V/Q_queue; :153-154 push), and an entry exists as soon as its frame is ENCODED
    (backbone + Encoding), before any attention.  So a round of W consecutive frames is:
        1. every rank encodes its own frame (all W backbones run concurrently) and exports the entry;
        2. ONE exchange: all-gather of the W entries (RCCL over xGMI; W broadcasts on backends without GPU all-gather);
        3. every rank walks the round in frame order: pushes the entries of the frames before its own, propagates its own
           frame (attention chain + head; this commits its own entry), pushes the entries after its own.
    Every rank's FIFO therefore goes through exactly the states of the sequential loop (test.py:45-53), outputs are
    bit-identical to one GPU serving the stream, and W frames finish in about one frame's latency.

    `stage` is duck-typed (the model classes of tdnet_amd.model implement it): encode(img, pos_id), propagate(labels=) -> out,
    cache_entry_numel_for(H, W) -> (nq, nk, nv), cache_export(q, k, v), cache_push(q, k, v), and optionally
    ensure_engine(H, W, device), called once before the first exchange so that a rank without a frame can still push entries.

    `frame_size` = (H, W) of the stream.  The exchange buffers are sized from it (pure arithmetic on the architecture), NOT from a
    live engine: a rank that owns no frame of a short first round (T < world) still knows the geometry and takes part in every
    collective, so no rank can fail alone and leave the others waiting in an all-gather.  It is required when world > 1 and
    checked in the constructor, i.e. identically on every rank and before any collective.
    """

    def __init__(self, stage, path_num, rank=None, world=None, device=None, frame_size=None):
        self.stage, self.P = stage, path_num
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.device = device
        if self.world > 1 and frame_size is None:
            raise ValueError("PathParallelStream: frame_size=(H, W) is required when world_size > 1")
        self.frame_size = None if frame_size is None else (int(frame_size[0]), int(frame_size[1]))
        self._buf = None

    def owner(self, t):
        return t % self.world

    def _buffers(self, frames=()):
        if self._buf is None:
            if self.frame_size is None:                               # world == 1: every frame is ours
                f = next(x for x in frames if x is not None)
                self.frame_size = (int(f.shape[-2]), int(f.shape[-1]))
            nq, nk, nv = self.stage.cache_entry_numel_for(*self.frame_size)
            self._sizes = (nq, nk, nv)
            # A rank that owns no frame of a short round (T < world) never encodes, yet it pushes its peers' entries: the stage must
            # have its handle (FIFO) from the geometry alone, on every rank, before the first exchange.
            if hasattr(self.stage, "ensure_engine") and self.device is not None and torch.device(self.device).type == "cuda":
                self.stage.ensure_engine(self.frame_size[0], self.frame_size[1], self.device)
            self._buf = torch.zeros(self.world, nq + nk + nv, dtype=torch.float32, device=self.device)
        return self._buf

    def _split(self, row):
        nq, nk, nv = self._sizes
        return row[:nq], row[nq:nq + nk], row[nq + nk:]

    def _exchange(self, buf, n_valid):
        """Row j of `buf` is rank j's entry; after the call every rank holds rows 0..n_valid-1."""
        if self.world == 1:
            return
        if dist.get_backend() == "nccl" and n_valid == self.world:
            dist.all_gather_into_tensor(buf.view(-1), buf[self.rank].clone())
        else:
            for j in range(n_valid):
                dist.broadcast(buf[j], src=j)

    def process(self, frames, labels=False, first_frame=0):
        """frames: consecutive frames of the stream (sequence of [1,3,H,W] device tensors, or None for frames this rank does
        not own); frames[i] is frame number first_frame + i of the stream (first_frame a multiple of world_size).
        Returns {i: output} for the frames this rank served."""
        T, W = len(frames), self.world
        if first_frame % W:
            raise ValueError("first_frame must be a multiple of world_size")
        outs = {}
        for r0 in range(0, T, W):
            n_valid = min(W, T - r0)
            mine = r0 + self.rank if self.rank < n_valid else None
            if mine is not None:
                self.stage.encode(frames[mine], pos_id=(first_frame + mine) % self.P)
            buf = self._buffers(frames)
            if mine is not None:
                self.stage.cache_export(*self._split(buf[self.rank]))
            self._exchange(buf, n_valid)
            for j in range(n_valid):
                if j == self.rank:
                    outs[mine] = self.stage.propagate(labels=labels)
                else:
                    self.stage.cache_push(*self._split(buf[j]))
        return outs


class FramePipelinedStream:
    """One video stream on ONE GPU with W (= 2) frames in flight: PathParallelStream's round with the ranks replaced by LANES -- W handles
    with the same weights on W HIP streams of one process, the exchange a pair of device-to-device copies.

    A frame of a small map leaves most of the chip idle (720x960 fp16: most launches have fewer workgroups than the chip has CUs, and a
    K loop is a chain of latencies), and what frame t + 1 needs from frame t is only its cache entry, which exists once frame t is
    ENCODED (td4_psp18.py:145-154).  So lane j of a round encodes frame r0 + j beside the other lane, exports the entry, then walks the
    round in frame order: pushes the entries of the frames before its own (waiting for their encode events), propagates its own frame,
    pushes the ones after.  Both FIFOs go through exactly the states of the sequential loop (test.py:45-53): the outputs are bit for bit
    those of one handle, only the order of work on the chip changes.  Throughput mode: a frame's LATENCY does not improve (it grows by
    what the other lane takes from it); at 1024x2048, where a frame fills the chip, there is nothing to gain.

    `stages`: W model instances (tdnet_amd.model) loaded with the same state_dict.  The rounds of consecutive process() calls chain
    without a host or device join in between when join=False; the caller's stream then has to wait (`join()`) before it reads outputs.
    Lane 0 IS the caller's current stream: feed a stream from one HIP stream, and join() before moving to another."""

    def __init__(self, stages, path_num, device, frame_size):
        self.stages, self.P, self.device = list(stages), path_num, torch.device(device)
        self.W = len(self.stages)
        if self.W < 1:
            raise ValueError("FramePipelinedStream: at least one stage")
        H, Wd = int(frame_size[0]), int(frame_size[1])
        self._sizes = self.stages[0].cache_entry_numel_for(H, Wd)
        for st in self.stages:
            st.ensure_engine(H, Wd, self.device)
        # [round parity][lane]: lane j re-exports into its row two rounds later, after waiting for the "round done" events the other
        # lanes recorded behind their reads of it
        self._buf = torch.zeros(2, self.W, sum(self._sizes), dtype=torch.float32, device=self.device)
        self._done = [[None] * self.W, [None] * self.W]
        self._lanes, self._lanes_for, self._round = None, None, 0

    @classmethod
    def from_model(cls, model, path_num, device, frame_size, lanes=2):
        """`lanes` lanes on ONE weight block: lane 0 is `model` itself, every further lane an instance of the same class whose handle is a
        tdnet_create_shared handle of model's (own workspace, K/Q/V FIFO and streams; no second copy of the packed weights, no second
        fold / pack / upload -- the reference's module is one set of parameters too).  model.share_weights_with documents the ownership."""
        H, Wd = int(frame_size[0]), int(frame_size[1])
        model.ensure_engine(H, Wd, device)
        stages = [model]
        for _ in range(lanes - 1):
            kw = dict(nclass=model.nclass, model_path=None, backbone=model.backbone, kernel_opts=model.kernel_opts, synthetic_seed=model.synthetic_seed)
            if getattr(model, "_model_id", None) != 1:
                kw["path_num"] = model.path_num
            stages.append(type(model)(**kw).eval().share_weights_with(model))
        return cls(stages, path_num, device, frame_size)

    def _split(self, row):
        nq, nk, nv = self._sizes
        return row[:nq], row[nq:nq + nk], row[nq + nk:]

    def _lane_streams(self, cur):
        if self._lanes_for != cur.cuda_stream:
            from .model._base import _TDNetBase
            lanes = [cur]
            for _ in range(self.W - 1):
                lanes.append(_TDNetBase._stream_beside(lanes, self.device, self.stages[0].engine.lib))
            self._lanes, self._lanes_for = lanes, cur.cuda_stream
        return self._lanes

    def join(self):
        """The caller's current stream waits for every lane (needed after process(..., join=False) before outputs are read on it)."""
        if self._lanes:
            cur = torch.cuda.current_stream(self.device)
            for s in self._lanes[1:]:
                cur.wait_stream(s)

    def reset(self):
        """Empty both FIFOs (a new clip) and forget the rounds in flight; the lanes are joined first."""
        self.join()
        torch.cuda.current_stream(self.device).synchronize()
        for st in self.stages:
            st.reset()
        self._done = [[None] * self.W, [None] * self.W]

    def process(self, frames, labels=False, first_frame=0, join=True):
        """frames: consecutive frames of the stream ([1,3,H,W] device tensors); frames[i] is frame first_frame + i.  Returns the list of
        outputs in frame order (logits, or int32 labels).  A frame that fails (wrong size, a kernel error) leaves the lanes' FIFOs in
        different states: the stream is reset() -- as after the last frame of a clip -- and the exception re-raised."""
        try:
            return self._process(frames, labels, first_frame, join)
        except Exception:
            try:
                self.reset()
            except Exception:                                         # the original error is the one to report
                pass
            raise

    def _process(self, frames, labels, first_frame, join):
        T, W = len(frames), self.W
        cur = torch.cuda.current_stream(self.device)
        lanes = self._lane_streams(cur)
        ready = torch.cuda.Event()
        ready.record(cur)                                             # the frames are the caller's: produced on its stream
        for s in lanes[1:]:
            s.wait_event(ready)
        outs = [None] * T
        for r0 in range(0, T, W):
            n = min(W, T - r0)
            par = self._round & 1
            buf, done = self._buf[par], self._done[par]
            self._round += 1
            enc = []
            for j in range(n):
                with torch.cuda.stream(lanes[j]):
                    for i in range(W):
                        if i != j and done[i] is not None:
                            lanes[j].wait_event(done[i])               # lane i has read this row (two rounds ago)
                    if j:
                        # the frame is the caller's tensor, read here on ANOTHER stream: tell the allocator (and any producer that recycles
                        # buffers by stream order, e.g. dataloader.DevicePrefetcher's `freed` events are on the caller's stream only)
                        if hasattr(frames[r0 + j], "record_stream"):    # (the schedule simulator of tests/ feeds frame NUMBERS)
                            frames[r0 + j].record_stream(lanes[j])
                    self.stages[j].encode(frames[r0 + j], pos_id=(first_frame + r0 + j) % self.P)
                    self.stages[j].cache_export(*self._split(buf[j]))
                    e = torch.cuda.Event()
                    e.record(lanes[j])
                    enc.append(e)
            for j in range(W):                                        # a lane without a frame in a short last round still takes every entry
                with torch.cuda.stream(lanes[j]):
                    for i in range(n):
                        if i == j:
                            outs[r0 + j] = self.stages[j].propagate(labels=labels)
                            if j:
                                outs[r0 + j].record_stream(cur)        # allocated on the lane's stream, consumed on the caller's
                        else:
                            lanes[j].wait_event(enc[i])
                            self.stages[j].cache_push(*self._split(buf[i]))
                    done[j] = torch.cuda.Event()
                    done[j].record(lanes[j])
        # (Inputs and join=False: lane 0 IS the caller's stream and waits for enc[i] of every other lane before it pushes that lane's entry, so
        # the caller's stream is already ordered behind every lane's only READ of its input frame -- a producer that recycles frame buffers in
        # the caller's stream order, like dataloader.DevicePrefetcher, cannot overwrite a frame a lane still reads.)
        if join:
            self.join()
        return outs

