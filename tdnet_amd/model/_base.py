"""Shared host-side mirror of the reference's stateful TDNet modules (Testing/model/pspnet/td4_psp18.py, td2_psp50.py).

Same constructor arguments, same `forward(img, pos_id)` contract, same failure behaviour (AssertionError on bad
path_num/backbone, RuntimeError on strict-load mismatches and on a LayerNorm/input-size mismatch), but the arithmetic
runs in libtdnet_hip.so (hand-written gfx950 kernels) through the C ABI of include/tdnet.h.  PyTorch supplies device
memory and the current HIP stream only.  There is no CPU path here: a CPU tensor raises.

Differences from the reference, on purpose:
  * a missing checkpoint file is an ERROR unless `synthetic_seed` is given (the reference prints and silently keeps
    random init, td4_psp18.py:239-240);
  * `reset()` empties the K/Q/V FIFO so a second clip can be fed (the reference has no reset);
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
        self.pretrained_mp_load()

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
        """Strict key/shape check against the reference inventory happens in the C library at finalize time."""
        self._state = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in state_dict.items()}
        self._engine = None
        return self

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
        h, w = arch.feat_size(H), arch.feat_size(W)
        sd = self._state
        if sd is None:
            if self.synthetic_seed is None:
                raise RuntimeError("no weights loaded: give model_path or synthetic_seed (the HIP path never runs on "
                                   "unspecified random init)")
            sd = weights.synth_state_dict(self.spec, h, w, self.synthetic_seed)
        if self._model_id == 1:
            sd = {k: v for k, v in sd.items()}
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
        self._engine, self._engine_key = eng, key
        return eng

    # ---- nn.Module surface used by Testing/test.py:40-41,53 ------------------------------------------------------
    def _check_frame(self, img, pos_id):
        if not torch.is_tensor(img) or img.dim() != 4 or img.shape[1] != 3:
            raise RuntimeError("expected an image tensor [1,3,H,W]")
        if img.device.type != "cuda":
            raise TdnetError("tdnet_amd runs on MI355X only: got a %s tensor (no CPU fallback)" % img.device.type)
        if img.shape[0] != 1:
            raise RuntimeError("batch size must be 1: the K/Q/V FIFO holds one video stream (test.py feeds [1,3,H,W])")
        if pos_id not in range(self.path_num):
            raise RuntimeError("pos_id must be t mod %d" % self.path_num)

    def forward(self, img, pos_id=0):
        self._check_frame(img, pos_id)
        img = img.contiguous().float()
        eng = self._get_engine(img)
        out = torch.empty((1, self.nclass, img.shape[2], img.shape[3]), device=img.device, dtype=torch.float32)
        eng.forward(img.data_ptr(), pos_id, out.data_ptr(), torch.cuda.current_stream(img.device).cuda_stream)
        return out

    def forward_labels(self, img, pos_id=0):
        """model(img,pos_id).max(1)[1] without materialising the full-resolution logits; int32 [1,H,W]."""
        self._check_frame(img, pos_id)
        img = img.contiguous().float()
        eng = self._get_engine(img)
        out = torch.empty((1, img.shape[2], img.shape[3]), device=img.device, dtype=torch.int32)
        eng.forward_labels(img.data_ptr(), pos_id, out.data_ptr(), torch.cuda.current_stream(img.device).cuda_stream)
        return out

    # ---- split frame + cache transport (path-parallel single stream: parallel.PathParallelStream) ------------------
    def encode(self, img, pos_id=0):
        """First half of forward(): backbone + pyramid slice + Encoding; the frame's cache entry is left pending."""
        self._check_frame(img, pos_id)
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
        if self._engine is not None:
            self._engine.reset()

    @property
    def engine(self):
        return self._engine
