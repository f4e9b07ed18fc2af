"""One pos_id cycle of a clip captured into a hipGraph and replayed: the "frame calls only enqueue" contract of include/tdnet.h, used.

A frame call (tdnet_forward / tdnet_forward_labels) never synchronises with the host and allocates nothing once the handle exists; what
changes from frame to frame is (i) the path (pos_id = t mod P) and (ii) which slots of the K/Q/V ring are read and written
(td4_psp18.py:123-134 as a ring: td_frame.h free_slot / fifo_commit).  Both repeat with period P frames in steady state (td4: four slots in
rotation, four paths; td2: two and two), so P consecutive frames captured on one stream are a graph that can be replayed for every later
cycle of the clip: the device-side ring keeps rotating exactly as the eager loop would rotate it, the host enqueues ONE graph launch per P
frames instead of P x 64..86 kernel launches (0.23-0.30 ms of host time per frame at 1024x2048: bench.py `host_launch_us_per_frame`).

The capture is plain PyTorch (torch.cuda.CUDAGraph = hipGraph on ROCm): the library sees an ordinary caller stream that happens to be
capturing -- its internal streams (cache-only attention chain, second row-parity chain) join the capture through the same events that
order them in eager mode.  The reference's loop (Testing/test.py:45-59) is the eager form; this is what a serving process would run.
"""
import torch


class GraphedClip:
    """P frames per replay.  Build it in STEADY STATE at the start of a pos_id cycle: after a multiple of P frames, FIFO full."""

    def __init__(self, model, H, W, device, labels=False):
        self.model, self.P, self.labels = model, model.path_num, bool(labels)
        dev = torch.device(device)
        eng = model.ensure_engine(int(H), int(W), dev)
        if eng.fifo_len() < self.P - 1:
            raise RuntimeError("GraphedClip: capture in steady state -- feed at least %d frames first (the warm-up frames take another "
                               "branch, td4_psp18.py:142-143)" % self.P)
        self.stream = torch.cuda.Stream(dev)
        self.xin = torch.zeros((self.P, 1, 3, int(H), int(W)), device=dev, dtype=torch.float32)
        eng.warmup(self.stream.cuda_stream)                            # the handle's one host-synchronising step, outside the capture
        self.graph = torch.cuda.CUDAGraph()
        fn = model.forward_labels if self.labels else model.forward
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream):
            self.out = [fn(self.xin[j], pos_id=j) for j in range(self.P)]
        self.launches_per_cycle = None

    def replay(self, frames):
        """frames: P tensors [1,3,H,W] on the device (frame t of the cycle at index t mod P).  Returns the P static output tensors
        (overwritten by the next replay: clone what must survive).  Stream-ordered on the caller's current stream."""
        if len(frames) != self.P:
            raise ValueError("GraphedClip.replay: expected %d frames" % self.P)
        for j, f in enumerate(frames):
            self.xin[j].copy_(f, non_blocking=True)
        self.graph.replay()
        return self.out
