"""Shared host-side mirror of the reference's stateful TDNet modules (Testing/model/pspnet/td4_psp18.py, td2_psp50.py).

Same constructor arguments, same `forward(img, pos_id)` contract, same failure behaviour (AssertionError on bad
path_num/backbone, RuntimeError on strict-load mismatches and on a LayerNorm/input-size mismatch), but the arithmetic
runs in libtdnet_hip.so (hand-written gfx950 kernels) through the C ABI of include/tdnet.h.  PyTorch supplies device
memory and the current HIP stream only.  There is no CPU path here: a CPU tensor raises.

Differences from the reference, on purpose:
  * a missing checkpoint file is an ERROR unless `synthetic_seed` is given (the reference prints and silently keeps
    random init, td4_psp18.py:239-240);
  * `reset()` empties the K/Q/V FIFO so a second clip can be fed (the reference has no reset);
  * a batch of N > 1 frames is N independent video streams in the reference too (every cached tensor, LayerNorm plane and softmax is
    per sample; td4_psp18.py:123-154 with [N, Lk, 64] queue entries): here sample i runs on its own handle (own FIFO), created the
    first time a batch that large arrives, and the odd samples on a second HIP stream beside the caller's, joined before
    forward() returns: frames of small maps leave CUs idle that another sample's kernels fill (720x960 fp16, two samples: 1445
    frames/s instead of 1100; 1024x2048: no change -- profiles/r04z_multi_clip_*).  Like the reference's queues, the batch size
    must not change while frames are cached;
  * `load_state_dict(strict=False)` drops unexpected keys and reports missing ones like nn.Module does, but a missing tensor has no
    "constructor initialisation" to fall back on here: it is taken from the seeded synthetic generator when `synthetic_seed` is given
    and is an error at the first frame otherwise;
  * td4 accepts all three backbones the reference's constructor accepts (td4_psp18.py:52-66), including the never-shipped
    resnet50 (d_model = d_v = 2048): its attention runs as four 512-channel launches.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import arch, weights
from ..engine import Engine
from .._capi import TdnetError


class _TDNetBase(nn.Module):
    _model_id = None       # 4 or 2
    _spec_name = None      # "td4" / "td2"

    def __init__(self, nclass=21, norm_layer=None, backbone="resnet18", dilated=True, aux=True, multi_grid=True,
                 path_num=None, model_path=None, synthetic_seed=None, kernel_opts=None):
        """kernel_opts (not in the reference): dict of include/tdnet.h tdnet_opts fields for THIS instance's handle, e.g.
        {"winograd": 0} for all-direct convs or {"precision": 1} for the fp16-MFMA mode; None = library defaults."""
        super().__init__()
        assert backbone == "resnet50" or backbone == "resnet34" or backbone == "resnet18"
        assert path_num == self._model_id
        if not (dilated and multi_grid):
            raise NotImplementedError("only the dilated, multi-grid backbone the reference ships is implemented")
        self.psp_path = model_path
        self.path_num = path_num
        self.nclass = nclass
        self.backbone = backbone
        self.synthetic_seed = synthetic_seed
        self.kernel_opts = dict(kernel_opts or {})
        self._pending_shape = None
        self.spec = arch.model_spec(self._spec_name, nclass, backbone)
        self._state = None
        self._engine = None
        self._engine_key = None
        self._init_batch_state()
        self.pretrained_mp_load()

    def _init_batch_state(self):
        self._extra_engines = []                                       # handles of the batch samples 1 .. N-1 (own FIFO each)
        self._extra_streams = []                                       # ... and the HIP streams their frames are enqueued on
        self._extra_streams_for = None                                 # the caller's stream they were placed against
        self._missing_keys = []
        self._batch = None                                             # batch size of the frames currently cached

    # ---- weights ---------------------------------------------------------------------------------------------
    def pretrained_mp_load(self):
        """td4_psp18.py:232-240, except that a missing file raises (see module docstring)."""
        if self.psp_path is not None:
            if os.path.isfile(self.psp_path):
                print("Loading pretrained model from '{}'".format(self.psp_path))
                self.load_state_dict(torch.load(self.psp_path, map_location="cpu"), strict=True)
            elif self.synthetic_seed is None:
                raise FileNotFoundError("No pretrained found at '{}' (pass synthetic_seed=... to run on seeded "
                                        "synthetic weights)".format(self.psp_path))

    def load_state_dict(self, state_dict, strict=True):
        """strict=True (td4_psp18.py:237): the key/shape check against the reference inventory happens in the C library at finalize
        time and raises RuntimeError.  strict=False: unexpected keys are dropped and missing ones recorded, as nn.Module does; the
        result is nn.Module's (missing_keys, unexpected_keys) tuple.  Tensors of the wrong size raise in both modes, like torch."""
        from torch.nn.modules.module import _IncompatibleKeys
        st = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in state_dict.items()}
        missing, unexpected = [], []
        if not strict:
            known = arch.state_dict_shapes(self.spec, 1, 1)                # the key inventory does not depend on the feature size
            unexpected = [k for k in st if k not in known]
            missing = [k for k in known if k not in st and not k.endswith("num_batches_tracked")
                       and not (k.startswith("pretrained") and (k.endswith(".fc.weight") or k.endswith(".fc.bias")))]
            st = {k: v for k, v in st.items() if k in known}
        self._state = st
        self._missing_keys = missing
        self._close_engines()
        return _IncompatibleKeys(missing, unexpected)

    def _close_engines(self):
        for e in [self._engine] + list(getattr(self, "_extra_engines", [])):
            if e is not None:
                e.close()
        self._engine, self._engine_key, self._extra_engines, self._batch = None, None, [], None

    def state_dict(self, *a, **k):
        """{name: CPU tensor} with the reference's keys, so torch.save(model.state_dict()) round-trips like the reference's
        checkpoints (td4_psp18.py:236-237); num_batches_tracked entries are int64 scalars as nn.BatchNorm2d registers them."""
        from collections import OrderedDict
        out = OrderedDict()
        for name, v in (self._state or {}).items():
            a_ = np.asarray(v)
            out[name] = torch.from_numpy(np.array(a_, dtype=np.int64 if name.endswith("num_batches_tracked") else np.float32))
        return out

    # ---- engine ----------------------------------------------------------------------------------------------
    def _get_engine(self, img):
        n, c, H, W = img.shape
        return self._engine_for(H, W, img.device.index or 0, n)

    def _engines_for_batch(self, img):
        """One handle per batch sample: sample 0 on the model's own handle, sample i > 0 on its own (same weights, own FIFO)."""
        n, c, H, W = img.shape
        first = self._engine_for(H, W, img.device.index or 0, n)
        if self._batch is not None and n != self._batch and first.fifo_len() > 0:
            # the reference's queues hold [N, Lk, .] tensors: a frame with another N fails in torch.bmm (transformer.py:133)
            raise RuntimeError("Expected batch size %d (the batch size of the cached frames), got %d: reset() the model before feeding "
                               "streams of another batch size" % (self._batch, n))
        self._batch = n
        while len(self._extra_engines) < n - 1:
            # the samples of a batch share one nn.Module's parameters in the reference (td4_psp18.py:216-229): the extra handles share
            # sample 0's weight block (include/tdnet.h tdnet_create_shared) -- own workspace + FIFO + streams, no second copy of the weights
            self._extra_engines.append(first.share())
        return [first] + self._extra_engines[:n - 1]

    def ensure_engine(self, H, W, device):
        """Build the handle (weights, workspace, FIFO) from the stream geometry alone, before any frame is seen.  A path-parallel rank
        that owns no frame of a short first round never calls encode(), yet it must accept its peers' cache entries
        (parallel.PathParallelStream calls this before the first exchange)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise TdnetError("tdnet_amd runs on MI355X only: got device %s (no CPU fallback)" % device)
        return self._engine_for(int(H), int(W), device.index or 0)

    def _engine_for(self, H, W, dev, n=1):
        key = (H, W, dev)
        if self._engine is not None and self._engine_key == key:
            return self._engine
        if self._engine is not None:
            raise RuntimeError("input size/device changed from %s to %s: one model instance serves one stream geometry"
                               % (self._engine_key, key))
        self._engine, self._engine_key = self._build_engine(H, W, dev, n), key
        return self._engine

    def _build_engine(self, H, W, dev, n=1):
        h, w = arch.feat_size(H), arch.feat_size(W)
        sd = self._state
        if sd is None:
            if self.synthetic_seed is None:
                raise RuntimeError("no weights loaded: give model_path or synthetic_seed (the HIP path never runs on "
                                   "unspecified random init)")
            sd = weights.synth_state_dict(self.spec, h, w, self.synthetic_seed)
        if self._missing_keys:                                           # load_state_dict(strict=False) left these without values
            if self.synthetic_seed is None:
                raise RuntimeError("load_state_dict(strict=False) left %d tensor(s) without values (first: %s); the reference would run them at "
                                   "their random initialisation, which this path refuses -- pass synthetic_seed=... to fill them from the "
                                   "seeded generator" % (len(self._missing_keys), self._missing_keys[0]))
            fill = weights.synth_state_dict(self.spec, h, w, self.synthetic_seed)
            sd = dict(sd)
            for k in self._missing_keys:
                sd[k] = fill[k]
        ln = sd.get("layer_norm1.ln.weight")
        if ln is not None and tuple(ln.shape) != (h, w):
            # same failure the reference raises from nn.LayerNorm (td4_psp18.py:107-110 hard-codes [97,193])
            raise RuntimeError("Given normalized_shape=%s, expected input with shape [*, %d, %d], but got input of size"
                               "[%d, %d, %d, %d]" % (list(ln.shape), ln.shape[0], ln.shape[1], n, self.spec.d_v, h, w))
        try:
            eng = Engine(self._model_id, int(self.backbone[6:]), self.nclass, H, W, dev, opts=self.kernel_opts)
            eng.load_state_dict(sd)
        except TdnetError as e:
            raise RuntimeError("Error(s) in loading state_dict for %s:\n\t%s" % (type(self).__name__, e))
        return eng

    def share_weights_with(self, owner):
        """Serve another stream of the SAME geometry from `owner`'s weight block instead of loading this instance's own copy: this
        model's handle becomes a tdnet_create_shared handle of owner's (own workspace, FIFO and streams; include/tdnet.h).  Used for
        the second lane of parallel.FramePipelinedStream and for clips sharing a GPU (bench.py --clips-per-gpu).  `owner` must have its
        handle (ensure_engine / a first frame); either model may be closed first (the block is reference-counted)."""
        if owner._engine is None:
            raise RuntimeError("share_weights_with(): the owner has no handle yet -- call owner.ensure_engine(H, W, device) first")
        self._close_engines()
        self._state = owner._state
        self._missing_keys = list(owner._missing_keys)
        self._engine, self._engine_key = owner._engine.share(), owner._engine_key
        return self

    # ---- nn.Module surface used by Testing/test.py:40-41,53 ------------------------------------------------------
    def _check_frame(self, img, pos_id):
        if not torch.is_tensor(img) or img.dim() != 4 or img.shape[1] != 3:
            raise RuntimeError("expected an image tensor [1,3,H,W]")
        if img.device.type != "cuda":
            raise TdnetError("tdnet_amd runs on MI355X only: got a %s tensor (no CPU fallback)" % img.device.type)
        if img.shape[0] < 1:
            raise RuntimeError("expected an image tensor [N,3,H,W] with N >= 1")
        if pos_id not in range(self.path_num):
            raise RuntimeError("pos_id must be t mod %d" % self.path_num)

    def _for_each_sample(self, img, call):
        """call(i, engine, raw_stream) for every batch sample: even samples on the caller's stream, odd ones on a second stream, which
        waits for the caller's (the input is ready) and is joined again before this returns -- so the caller's stream stays the only
        one the caller has to order against, as with a single handle.  N = 1 (test.py:46-53): one handle, one call, no events."""
        engines = self._engines_for_batch(img)
        cur = torch.cuda.current_stream(img.device)
        if len(engines) == 1:
            call(0, engines[0], cur.cuda_stream)
            return
        if self._extra_streams_for != cur.cuda_stream:                 # placed against the caller's stream: another caller stream, again
            self._extra_streams, self._extra_streams_for = [], cur.cuda_stream
        # TWO lanes whatever N: even samples on the caller's stream, odd ones on one extra stream.  A third concurrent frame gains
        # nothing and can cost a lot (720x960 fp16, frames/s in total: 1 sample 1100, 2: 1445, 3 on three streams: 1280 with two
        # hardware queues, 850 with three -- the process then owns more busy queues than stay resident; profiles/r04z_*batched*)
        if not self._extra_streams:
            self._extra_streams.append(self._stream_beside([cur], img.device, engines[0].lib))
        side = self._extra_streams[0]
        ready = torch.cuda.Event()
        ready.record(cur)
        side.wait_event(ready)
        try:
            for i, eng in enumerate(engines):
                call(i, eng, (side if i & 1 else cur).cuda_stream)
        except BaseException:
            # a sample failed (TdnetError from the C library): the samples before it have committed their frame, the ones after it
            # have not -- the per-sample FIFOs are out of step.  Like FramePipelinedStream.process: every stream starts over.
            for eng in engines:
                eng.reset()
            self._batch = None
            raise
        finally:
            cur.wait_stream(side)                                      # whatever happened, the caller's stream is ordered behind the side lane:
                                                                       # `out` and the temporary input were allocated on it and may be freed now

    @staticmethod
    def _stream_beside(taken, device, lib):
        """A stream whose kernels really run BESIDE those of `taken`: HIP deals streams onto a small pool of hardware queues and two
        streams on one queue serialise (which pair collides depends on how many streams the process has created -- with three queues
        the second sample's first candidate lands on the caller's queue: 1066 instead of 1430 frames/s for two 720x960 clips,
        profiles/r04z_hw_queues_*).  Candidates from torch's pool are tried against every taken stream with the library's spin-pair test
        (include/tdnet.h tdnet_streams_share_queue); with more samples than queues the last candidate is used as it is."""
        import ctypes
        s = None
        for _ in range(8):
            s = torch.cuda.Stream(device)
            shared = ctypes.c_int(0)
            for t in taken:
                lib.check(lib.tdnet_streams_share_queue(t.cuda_stream, s.cuda_stream, ctypes.byref(shared)))
                if shared.value:
                    break
            if not shared.value:
                break
        return s

    def forward(self, img, pos_id=0):
        self._check_frame(img, pos_id)
        img = img.contiguous().float()
        N = img.shape[0]
        out = torch.empty((N, self.nclass, img.shape[2], img.shape[3]), device=img.device, dtype=torch.float32)
        self._for_each_sample(img, lambda i, eng, s: eng.forward(img[i].data_ptr(), pos_id, out[i].data_ptr(), s))
        return out

    def forward_labels(self, img, pos_id=0):
        """model(img,pos_id).max(1)[1] without materialising the full-resolution logits; int32 [1,H,W]."""
        self._check_frame(img, pos_id)
        img = img.contiguous().float()
        N = img.shape[0]
        out = torch.empty((N, img.shape[2], img.shape[3]), device=img.device, dtype=torch.int32)
        self._for_each_sample(img, lambda i, eng, s: eng.forward_labels(img[i].data_ptr(), pos_id, out[i].data_ptr(), s))
        return out

    # ---- split frame + cache transport (path-parallel single stream: parallel.PathParallelStream) ------------------
    def encode(self, img, pos_id=0):
        """First half of forward(): backbone + pyramid slice + Encoding; the frame's cache entry is left pending."""
        self._check_frame(img, pos_id)
        if img.shape[0] != 1:
            raise RuntimeError("encode(): the split frame serves ONE stream (batch size 1)")
        img = img.contiguous().float()
        eng = self._get_engine(img)
        self._pending_shape = (img.shape[2], img.shape[3], img.device)
        eng.encode(img.data_ptr(), pos_id, torch.cuda.current_stream(img.device).cuda_stream)

    def propagate(self, labels=False):
        """Second half of forward() for the pending frame, against the FIFO as it stands; returns logits (or int32 labels)."""
        if self._engine is None or self._pending_shape is None:
            raise RuntimeError("propagate(): no encoded frame is pending (call encode(img, pos_id) first)")
        H, W, dev = self._pending_shape
        self._pending_shape = None
        s = torch.cuda.current_stream(dev).cuda_stream
        if labels:
            out = torch.empty((1, H, W), device=dev, dtype=torch.int32)
            self._engine.propagate_labels(out.data_ptr(), s)
        else:
            out = torch.empty((1, self.nclass, H, W), device=dev, dtype=torch.float32)
            self._engine.propagate(out.data_ptr(), s)
        return out

    def cache_entry_numel(self):
        """(q, k, v) element counts of one cache entry: [Lk,64], [Lk,64], [Lk,d_v]."""
        if self._engine is None:
            raise RuntimeError("cache_entry_numel(): the geometry is known once a frame has been encoded; use cache_entry_numel_for(H, W) before")
        lk, dk, dv = self._engine.cache_dims()
        return lk * dk, lk * dk, lk * dv

    def cache_entry_numel_for(self, H, W):
        """The same from the input size alone (arch): lets every rank of a path-parallel group size its buffers before any frame."""
        lk = arch.key_size(arch.feat_size(H)) * arch.key_size(arch.feat_size(W))
        return lk * self.spec.d_k, lk * self.spec.d_k, lk * self.spec.d_v

    def cache_export(self, q, k, v):
        """Copy the pending frame's cache entry into the given contiguous fp32 CUDA tensors."""
        self._engine.cache_export(q.data_ptr(), k.data_ptr(), v.data_ptr(), torch.cuda.current_stream(q.device).cuda_stream)

    def cache_push(self, q, k, v):
        """Append a peer's cache entry to the FIFO (contiguous fp32 CUDA tensors)."""
        if self._engine is None:
            raise RuntimeError("cache_push(): no handle yet -- call ensure_engine(H, W, device) (or encode a frame) first")
        self._engine.cache_push(q.data_ptr(), k.data_ptr(), v.data_ptr(), torch.cuda.current_stream(q.device).cuda_stream)

    def reset(self):
        for e in [self._engine] + list(self._extra_engines):
            if e is not None:
                e.reset()
        self._batch = None
        self._pending_shape = None                                     # a frame encoded but not propagated is dropped with the FIFO (tdnet_reset)

    @property
    def engine(self):
        return self._engine
