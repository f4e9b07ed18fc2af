"""Thin, torch-free owner of one tdnet handle (one video stream on one GPU).

Pointers are plain integers: the model classes pass `tensor.data_ptr()` of torch-ROCm tensors; PyTorch is plumbing
for device memory and streams only.
"""
import ctypes

import numpy as np

from . import _capi


def _ptr(a):
    """int address of a numpy array / integer pointer / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    return a.ctypes.data


class Engine:
    def __init__(self, model, backbone, nclass, height, width, device=0, lib=None, opts=None, _shared_from=None):
        """opts: None (library defaults) or a dict of tdnet_opts fields (winograd=, precision=, pipeline=, ...): per handle."""
        self.lib = lib or _capi.lib()
        self.cfg = _capi.TdnetCfg(model, backbone, nclass, height, width, device)
        h = ctypes.c_void_p()
        if _shared_from is not None:                                   # a further handle on the same weight block (tdnet_create_shared)
            self.lib.check(self.lib.tdnet_create_shared(_shared_from.h, None, ctypes.byref(h)))
            self.h = h
            self.finalized = True
            return
        o = self.lib.opts(**(opts or {}))
        self.lib.check(self.lib.tdnet_create_opts(ctypes.byref(self.cfg), ctypes.byref(o), ctypes.byref(h)))
        self.h = h
        self.finalized = False

    def share(self):
        """A new Engine on THIS engine's weights (one copy of the packed weights in HBM): own workspace, own K/Q/V FIFO, own streams.
        The weight block is reference-counted in the library: either engine may be closed first."""
        if not self.finalized:
            raise _capi.TdnetError("share(): load_state_dict() first")
        c = self.cfg
        return Engine(c.model, c.backbone, c.nclass, c.height, c.width, c.device, lib=self.lib, _shared_from=self)

    def warmup(self, stream=None):
        """The one host-synchronising step of a handle (placement of its internal streams against `stream`), done now instead of
        inside the first frame (include/tdnet.h "Conventions")."""
        self.lib.check(self.lib.tdnet_warmup(self.h, stream))

    def memory_bytes(self):
        """(bytes of the shared weight block, bytes of this handle alone, handles sharing the block)."""
        w, m = ctypes.c_size_t(), ctypes.c_size_t()
        n = self.lib.check(self.lib.tdnet_memory_bytes(self.h, ctypes.byref(w), ctypes.byref(m)))
        return w.value, m.value, n

    def last_launch_count(self):
        return self.lib.tdnet_last_launch_count(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tdnet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights: strict, like load_state_dict(strict=True) (td4_psp18.py:236-237)
    def load_state_dict(self, sd):
        for name, v in sd.items():
            a = np.ascontiguousarray(np.asarray(v, dtype=np.float32)).reshape(-1)
            if a.size == 0:
                a = np.zeros(1, np.float32)
            self.lib.check(self.lib.tdnet_set_weight(self.h, name.encode(), a.ctypes.data, a.size))
        self.lib.check(self.lib.tdnet_finalize_weights(self.h))
        self.finalized = True

    def forward(self, img_ptr, pos_id, logits_ptr, stream=None):
        self.lib.check(self.lib.tdnet_forward(self.h, _ptr(img_ptr), int(pos_id), _ptr(logits_ptr), stream))

    def forward_labels(self, img_ptr, pos_id, labels_ptr, stream=None):
        self.lib.check(self.lib.tdnet_forward_labels(self.h, _ptr(img_ptr), int(pos_id), _ptr(labels_ptr), stream))

    # ---- split frame + cache transport (path-parallel single stream; include/tdnet.h) ----
    def encode(self, img_ptr, pos_id, stream=None):
        self.lib.check(self.lib.tdnet_encode(self.h, _ptr(img_ptr), int(pos_id), stream))

    def propagate(self, logits_ptr, stream=None):
        self.lib.check(self.lib.tdnet_propagate(self.h, _ptr(logits_ptr), stream))

    def propagate_labels(self, labels_ptr, stream=None):
        self.lib.check(self.lib.tdnet_propagate_labels(self.h, _ptr(labels_ptr), stream))

    def cache_dims(self):
        import ctypes
        lk, dk, dv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.lib.check(self.lib.tdnet_cache_dims(self.h, ctypes.byref(lk), ctypes.byref(dk), ctypes.byref(dv)))
        return lk.value, dk.value, dv.value

    def cache_export(self, q_ptr, k_ptr, v_ptr, stream=None):
        self.lib.check(self.lib.tdnet_cache_export(self.h, _ptr(q_ptr), _ptr(k_ptr), _ptr(v_ptr), stream))

    def cache_push(self, q_ptr, k_ptr, v_ptr, stream=None):
        self.lib.check(self.lib.tdnet_cache_push(self.h, _ptr(q_ptr), _ptr(k_ptr), _ptr(v_ptr), stream))

    def argmax(self, logits_ptr, labels_ptr, stream=None):
        self.lib.check(self.lib.tdnet_argmax(self.h, _ptr(logits_ptr), _ptr(labels_ptr), stream))

    def reset(self):
        self.lib.check(self.lib.tdnet_reset(self.h))

    def fifo_len(self):
        return self.lib.tdnet_fifo_len(self.h)

    def stage(self, name, shape):
        out = np.empty(int(np.prod(shape)), np.float32)
        n = self.lib.check(self.lib.tdnet_get_stage(self.h, name.encode(), out.ctypes.data, out.size))
        assert n == out.size, (name, n, out.size)
        return out.reshape(shape)

    def opts(self):
        o = _capi.TdnetOpts()
        self.lib.check(self.lib.tdnet_get_opts(self.h, ctypes.byref(o)))
        return o.as_dict()

    def flops_per_frame(self):
        return self.lib.tdnet_flops_per_frame(self.h)

    def set_profiling(self, on):
        self.lib.check(self.lib.tdnet_set_profiling(self.h, int(on)))

    def last(self, which):
        """(ms, algorithmic flop, launches) of a kernel family in the last forward; see include/tdnet.h."""
        return (self.lib.tdnet_last_ms(self.h, which), self.lib.tdnet_last_flops(self.h, which),
                self.lib.tdnet_last_launches(self.h, which))
