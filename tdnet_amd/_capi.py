"""ctypes binding of include/tdnet.h (libtdnet_hip.so) and, for tests / tools only, of include/tdnet_test.h (libtdnet_hip_test.so).

The HIP library is the product: importing a model fails loudly when it is missing -- there is no CPU or
torch fallback anywhere in this package.  The product library exports the C ABI of include/tdnet.h and nothing else; the single-operator entry
points and probes the tests use live in a second library built from the same sources (tdnet_amd/build.py), loaded by `test_lib()`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libtdnet_hip.so")
TEST_LIB = os.path.join(_HERE, "lib", "libtdnet_hip_test.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_void_p = ctypes.c_void_p


class TdnetCfg(ctypes.Structure):
    _fields_ = [("model", ctypes.c_int32), ("backbone", ctypes.c_int32), ("nclass", ctypes.c_int32),
                ("height", ctypes.c_int32), ("width", ctypes.c_int32), ("device", ctypes.c_int32)]


class TdnetOpts(ctypes.Structure):
    """include/tdnet.h tdnet_opts: per-handle kernel configuration (nothing in the library is process-wide)."""
    _fields_ = [("winograd", ctypes.c_int32), ("precision", ctypes.c_int32), ("pipeline", ctypes.c_int32),
                ("gemm_persistent", ctypes.c_int32), ("reserved0", ctypes.c_int32), ("attention", ctypes.c_int32),
                ("fusion", ctypes.c_int32), ("overlap", ctypes.c_int32), ("reserved", ctypes.c_int32 * 8)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class TdnetError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol include/tdnet.h declares
WINOGRAD_DEFAULT = 3          # include/tdnet.h TDNET_WINOGRAD_DEFAULT: F(4x4,3x3) for the wide stride-1 3x3 convs
c_opts_p = ctypes.POINTER(TdnetOpts)

SYMBOLS = {
    "tdnet_opts_default": (None, [c_opts_p]),
    "tdnet_create": (ctypes.c_int, [ctypes.POINTER(TdnetCfg), ctypes.POINTER(c_void_p)]),
    "tdnet_create_opts": (ctypes.c_int, [ctypes.POINTER(TdnetCfg), c_opts_p, ctypes.POINTER(c_void_p)]),
    "tdnet_get_opts": (ctypes.c_int, [c_void_p, c_opts_p]),
    "tdnet_destroy": (None, [c_void_p]),
    "tdnet_set_weight": (ctypes.c_int, [c_void_p, ctypes.c_char_p, c_void_p, ctypes.c_size_t]),
    "tdnet_finalize_weights": (ctypes.c_int, [c_void_p]),
    "tdnet_create_shared": (ctypes.c_int, [c_void_p, c_opts_p, ctypes.POINTER(c_void_p)]),
    "tdnet_warmup": (ctypes.c_int, [c_void_p, c_void_p]),
    "tdnet_memory_bytes": (ctypes.c_int, [c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "tdnet_last_launch_count": (ctypes.c_int, [c_void_p]),
    "tdnet_streams_share_queue": (ctypes.c_int, [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "tdnet_forward": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "tdnet_argmax": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "tdnet_forward_labels": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int, c_void_p, c_void_p]),
    "tdnet_reset": (ctypes.c_int, [c_void_p]),
    "tdnet_fifo_len": (ctypes.c_int, [c_void_p]),
    "tdnet_encode": (ctypes.c_int, [c_void_p, c_void_p, ctypes.c_int, c_void_p]),
    "tdnet_propagate": (ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    "tdnet_propagate_labels": (ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    "tdnet_cache_dims": (ctypes.c_int, [c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "tdnet_cache_export": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tdnet_cache_push": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tdnet_get_stage": (ctypes.c_long, [c_void_p, ctypes.c_char_p, c_void_p, ctypes.c_size_t]),
    "tdnet_flops_per_frame": (ctypes.c_double, [c_void_p]),
    "tdnet_set_profiling": (ctypes.c_int, [c_void_p, ctypes.c_int]),
    "tdnet_last_ms": (ctypes.c_double, [c_void_p, ctypes.c_int]),
    "tdnet_last_flops": (ctypes.c_double, [c_void_p, ctypes.c_int]),
    "tdnet_last_launches": (ctypes.c_double, [c_void_p, ctypes.c_int]),
    "tdnet_last_error": (ctypes.c_char_p, []),
    "tdnet_version": (ctypes.c_char_p, []),
}


# include/tdnet_test.h: libtdnet_hip_test.so (and the emulator library) only
TEST_SYMBOLS = {
    "tdnet_bench_mfma_peak": (ctypes.c_double, [ctypes.c_int, ctypes.c_int, c_void_p]),
    "tdnet_bench_conv": (ctypes.c_double, [ctypes.c_int] * 9 + [c_opts_p, c_void_p]),
    "tdnet_op_conv2d": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p, ctypes.c_int, c_opts_p, ctypes.c_int,
                                       c_void_p, c_void_p]),
    "tdnet_op_conv2d_f16io": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, c_void_p, ctypes.c_int, ctypes.c_int,
                                             c_void_p, c_void_p]),
    "tdnet_op_stem": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, c_opts_p, c_void_p, c_void_p]),
    "tdnet_op_attention": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tdnet_op_layernorm_hw": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tdnet_op_ppm": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, ctypes.c_int, ctypes.c_int,
                                    c_void_p, c_void_p]),
    "tdnet_op_upsample": (ctypes.c_int, [c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         c_void_p, c_void_p]),
}


class Lib:
    """A loaded libtdnet shared object with typed entry points; `check()` turns return codes into TdnetError.  test_symbols: also bind the
    entry points of include/tdnet_test.h (the tests' library and the emulator build carry them, the product library does not)."""

    def __init__(self, path=DEFAULT_LIB, test_symbols=False):
        if not os.path.exists(path):
            raise TdnetError("HIP library not found: %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        if path in (DEFAULT_LIB, TEST_LIB):
            # PyTorch-ROCm bundles its own HIP runtime under the same SONAME; it must be the one already loaded when this
            # library's libamdhip64 dependency is resolved, or the process ends up with two runtimes and hipSetDevice reports
            # "no ROCm-capable device" (seen on the GPU box when the library was loaded before `import torch`).
            import torch  # noqa: F401
        self.dll = ctypes.CDLL(path)
        for name, (res, args) in list(SYMBOLS.items()) + (list(TEST_SYMBOLS.items()) if test_symbols else []):
            fn = getattr(self.dll, name)              # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def opts(self, **kw):
        """tdnet_opts with the library defaults, overridden by keyword (winograd=0, precision=1, ...)."""
        o = TdnetOpts()
        self.tdnet_opts_default(ctypes.byref(o))
        for k, v in kw.items():
            if k not in dict(TdnetOpts._fields_) or k.startswith("reserved"):
                raise TypeError("unknown tdnet_opts field %r" % k)
            if v is not None:
                setattr(o, k, int(v))
        return o

    def check(self, rc):
        if rc is not None and rc < 0:
            raise TdnetError(self.tdnet_last_error().decode())
        return rc


_default = None
_test = None


def lib():
    """The product library (in-tree libtdnet_hip.so), loaded once."""
    global _default
    if _default is None:
        _default = Lib(DEFAULT_LIB)
    return _default


def test_lib():
    """tests / tools only: the superset library with the single-operator entry points and probes of include/tdnet_test.h."""
    global _test
    if _test is None:
        _test = Lib(TEST_LIB, test_symbols=True)
    return _test
