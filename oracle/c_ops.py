"""CPU ORACLE (test infrastructure, NOT product code): ctypes binding of oracle/ops_c.c -- the plain-C restatement of the
L0 operators under oracle/tdnet_ref.py's graph.  `tdnet_ref.set_ops(COps)` runs the whole oracle without a single PyTorch
kernel (torch tensors are used as containers only: reshape / slice / cat / elementwise fp32 arithmetic).

Only tests/ may import this file.  build() compiles the C file with gcc into oracle/_build/ (git-ignored; it travels to the GPU
box with the snapshot like the other built libraries).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ops_c.c")
OUT = os.path.join(HERE, "_build", "libtdnet_oracle_ops.so")
_lib = None


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", SRC, "-o", OUT, "-lm"], check=True)
    return OUT


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(t):
    """contiguous fp32 numpy view of a torch tensor"""
    return np.ascontiguousarray(t.detach().to(torch.float32).numpy())


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class COps:
    """The subset of torch / torch.nn.functional that oracle/tdnet_ref.py calls, on the C restatement (N = 1 images)."""

    @staticmethod
    def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        assert groups == 1 and x.shape[0] == 1
        s, p, d = _pair(stride)[0], _pair(padding)[0], _pair(dilation)[0]
        xa, wa = _f(x), _f(w)
        _, C, H, W = xa.shape
        O, _, KS, _ = wa.shape
        Ho = (H + 2 * p - d * (KS - 1) - 1) // s + 1
        Wo = (W + 2 * p - d * (KS - 1) - 1) // s + 1
        y = np.empty((1, O, Ho, Wo), np.float32)
        ba = _f(bias) if bias is not None else None
        lib().tdc_conv2d(_p(xa), C, H, W, _p(wa), _p(ba) if ba is not None else None, O, KS, s, p, d, _p(y), Ho, Wo)
        return torch.from_numpy(y)

    @staticmethod
    def relu(x):
        return torch.where(x > 0, x, torch.zeros_like(x))

    @staticmethod
    def leaky_relu(x, slope=0.01):
        return torch.where(x > 0, x, x * slope)

    @staticmethod
    def max_pool2d(x, k, stride, pad):
        xa = _f(x)
        _, C, H, W = xa.shape
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = np.empty((1, C, Ho, Wo), np.float32)
        lib().tdc_max_pool2d(_p(xa), C, H, W, k, stride, pad, _p(y), Ho, Wo)
        return torch.from_numpy(y)

    @staticmethod
    def adaptive_avg_pool2d(x, o):
        xa = _f(x)
        _, C, H, W = xa.shape
        y = np.empty((1, C, o, o), np.float32)
        lib().tdc_adaptive_avg_pool2d(_p(xa), C, H, W, o, _p(y))
        return torch.from_numpy(y)

    @staticmethod
    def interpolate(x, size, mode="bilinear", align_corners=True):
        assert mode == "bilinear" and align_corners
        xa = _f(x)
        _, C, h, w = xa.shape
        H, W = size
        y = np.empty((1, C, H, W), np.float32)
        lib().tdc_bilinear_ac(_p(xa), C, h, w, _p(y), H, W)
        return torch.from_numpy(y)

    @staticmethod
    def layer_norm(x, shape, weight, bias, eps):
        xa, g, b = _f(x), _f(weight), _f(bias)
        _, C, H, W = xa.shape
        assert tuple(shape) == (H, W)
        y = np.empty_like(xa)
        lib().tdc_layer_norm_plane(_p(xa), C, H * W, _p(g), _p(b), ctypes.c_double(eps), _p(y))
        return torch.from_numpy(y)

    @staticmethod
    def bmm(a, b):
        """[1,M,K] x [1,K,N]; a transposed view of b (q k^T, transformer.py:132) goes to the NT kernel"""
        assert a.shape[0] == 1 and b.shape[0] == 1
        aa = _f(a[0])
        M, K = aa.shape
        N = b.shape[2]
        c = np.empty((1, M, N), np.float32)
        if b[0].t().is_contiguous() and not b[0].is_contiguous():
            lib().tdc_bmm_nt(_p(aa), _p(_f(b[0].t())), _p(c), M, K, N)
        else:
            lib().tdc_bmm_nn(_p(aa), _p(_f(b[0])), _p(c), M, K, N)
        return torch.from_numpy(c)

    @staticmethod
    def matmul(a, b):
        """[1,M,K] x [K,N]"""
        return COps.bmm(a, b[None])

    @staticmethod
    def softmax(x, dim):
        assert dim == x.dim() - 1
        xa = _f(x)
        y = np.empty_like(xa)
        lib().tdc_softmax_lastdim(_p(xa), _p(y), int(np.prod(xa.shape[:-1])), xa.shape[-1])
        return torch.from_numpy(y)
