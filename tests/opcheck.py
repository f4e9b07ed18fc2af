"""Operator-level parity checks shared by the emulator tests (CPU, numpy "HBM") and the GPU tests (torch-ROCm HBM).
Each check builds seeded inputs, runs ONE C-ABI op, and compares with the same op in plain PyTorch fp32 on CPU."""
import numpy as np
import torch
import torch.nn.functional as F


class NumpyMem:
    """'device' memory for the emulator: host arrays."""
    def put(self, a):
        return np.ascontiguousarray(a, dtype=np.float32)

    def empty(self, shape):
        return np.full(shape, 7e7, np.float32)

    def ptr(self, a):
        return None if a is None else a.ctypes.data

    def get(self, a):
        return a

    stream = None


class TorchMem:
    """device memory on cuda:0 through torch-ROCm (plumbing only)."""
    def put(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

    def empty(self, shape):
        return torch.full(tuple(shape), 7e7, dtype=torch.float32, device="cuda")

    def ptr(self, a):
        return None if a is None else a.data_ptr()

    def get(self, a):
        torch.cuda.synchronize()
        return a.cpu().numpy()

    @property
    def stream(self):
        return torch.cuda.current_stream().cuda_stream


def conv(lib, mem, H, W, Cin, Cout, KS, stride, dil, act, resid, tile=None, seed=0, tol=1e-4, opts=None):
    """opts: dict of tdnet_opts fields for this one call (e.g. {"winograd": 0}); None = library defaults."""
    import ctypes
    g = np.random.default_rng(seed)
    x = g.standard_normal((H, W, Cin)).astype(np.float32)
    w = (g.standard_normal((Cout, Cin, KS, KS)) * (1.0 / np.sqrt(Cin * KS * KS))).astype(np.float32)
    b = g.standard_normal(Cout).astype(np.float32)
    pad = dil * (KS // 2)
    ref = F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(w), torch.from_numpy(b), stride, pad, dil)
    Ho, Wo = ref.shape[-2:]
    r = None
    if resid:
        r = g.standard_normal((Ho, Wo, Cout)).astype(np.float32)
        ref = ref + torch.from_numpy(r).permute(2, 0, 1)[None]
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.leaky_relu(ref, 0.01)
    dx, dr, out = mem.put(x), (mem.put(r) if resid else None), mem.empty((Ho, Wo, Cout))
    o = lib.opts(**(opts or {}))
    rc = lib.tdnet_op_conv2d(mem.ptr(dx), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, KS, stride, dil, mem.ptr(dr), act,
                             ctypes.byref(o), -1 if tile is None else tile, mem.ptr(out), mem.stream)
    lib.check(rc)
    err = float(np.abs(mem.get(out) - ref[0].permute(1, 2, 0).numpy()).max())
    assert err <= tol, ("conv", H, W, Cin, Cout, KS, stride, dil, act, resid, tile, err)
    return err


def conv_f16io(lib, mem, H, W, Cin, Cout, KS, stride, dil, act, resid, tile=None, seed=0, tol=4e-3, want_out=False):
    """The fp16-storage conv of tdnet_opts.precision = 1 (input / residual / output maps fp16 in HBM, fp16 MFMA, fp32 accumulate)
    against an fp64 evaluation on the fp16-rounded operands; what is left is fp32 summation order and the output's own rounding to
    fp16 (2^-11 relative)."""
    g = np.random.default_rng(seed)
    x = g.standard_normal((H, W, Cin)).astype(np.float32)
    w = (g.standard_normal((Cout, Cin, KS, KS)) / np.sqrt(Cin * KS * KS)).astype(np.float32)
    b = g.standard_normal(Cout).astype(np.float32)
    pad = dil * (KS // 2)
    h = lambda a: torch.from_numpy(a).half().double()
    ref = F.conv2d(h(x).permute(2, 0, 1)[None], h(w), torch.from_numpy(b).double(), stride, pad, dil)
    Ho, Wo = ref.shape[-2:]
    r = None
    if resid:
        r = g.standard_normal((Ho, Wo, Cout)).astype(np.float32)
        ref = ref + h(r).permute(2, 0, 1)[None]
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.leaky_relu(ref, 0.01)
    ref = ref[0].permute(1, 2, 0).float().numpy()
    dx, dr, out = mem.put(x), (mem.put(r) if resid else None), mem.empty((Ho, Wo, Cout))
    lib.check(lib.tdnet_op_conv2d_f16io(mem.ptr(dx), H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, KS, stride, dil, mem.ptr(dr), act,
                                        -1 if tile is None else tile, mem.ptr(out), mem.stream))
    got = mem.get(out)
    err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
    assert err <= tol, ("conv_f16io", H, W, Cin, Cout, KS, stride, dil, act, resid, tile, err)
    return (err, got) if want_out else err


def stem(lib, mem, H, W, seed=0, tol=1e-4, opts=None):
    import ctypes
    g = np.random.default_rng(seed)
    img = g.standard_normal((3, H, W)).astype(np.float32)
    w = (g.standard_normal((64, 3, 7, 7)) * 0.1).astype(np.float32)
    b = g.standard_normal(64).astype(np.float32)
    ref = F.max_pool2d(F.relu(F.conv2d(torch.from_numpy(img)[None], torch.from_numpy(w), torch.from_numpy(b), 2, 3)), 3, 2, 1)
    ref = ref[0].permute(1, 2, 0).numpy()
    di, out = mem.put(img), mem.empty(ref.shape)
    lib.check(lib.tdnet_op_stem(mem.ptr(di), H, W, w.ctypes.data, b.ctypes.data, ctypes.byref(lib.opts(**(opts or {}))), mem.ptr(out), mem.stream))
    err = float(np.abs(mem.get(out) - ref).max())
    assert err <= tol, ("stem", H, W, err)
    return err


def attention(lib, mem, Lq, Lk, DV, bias=True, resid=True, seed=0, tol=1e-4, qk_scale=1.0, spike=False, online=0, ln=False, ramp=False):
    """online: the single-pass schedule; ln: also check the plane LayerNorm computed from the epilogue's strip statistics;
    ramp: keys sorted by growing score for every query, so the online reference has to move again and again."""
    g = np.random.default_rng(seed)
    q = (qk_scale * g.standard_normal((Lq, 64))).astype(np.float32)
    k = (qk_scale * g.standard_normal((Lk, 64))).astype(np.float32)
    if spike:                                  # one key dominating one query row: exercises the max subtraction
        k[Lk // 2] = 6.0 * q[Lq // 3] / max(1e-6, float(np.linalg.norm(q[Lq // 3]))) * 8.0
    if ramp:                                   # k_j = (j / Lk) * 40 * u with q . u > 0 for all q: scores grow along the key axis by ~100 log2 units
        u = np.ones(64, np.float32) / 8.0
        q = np.abs(q) + 0.5
        k = (np.arange(Lk, dtype=np.float32)[:, None] / max(1, Lk - 1)) * 40.0 * u[None, :] + 0.1 * k
    v = g.standard_normal((Lk, DV)).astype(np.float32)
    b = g.standard_normal(DV).astype(np.float32)
    r = g.standard_normal((Lq, DV)).astype(np.float32)
    ref = torch.softmax(torch.from_numpy(q).double() @ torch.from_numpy(k).double().T / 8.0, 1) @ torch.from_numpy(v).double()
    if bias:
        ref = ref + torch.from_numpy(b)
    if resid:
        ref = ref + torch.from_numpy(r)
    dq, dk, dv_, db, dr = mem.put(q), mem.put(k), mem.put(v), mem.put(b), mem.put(r)
    out = mem.empty((Lq, DV))
    gg = g.uniform(0.5, 1.5, Lq).astype(np.float32)
    bb = g.standard_normal(Lq).astype(np.float32)
    dg, dbb, lnout = (mem.put(gg), mem.put(bb), mem.empty((Lq, DV))) if ln else (None, None, None)
    lib.check(lib.tdnet_op_attention(mem.ptr(dq), mem.ptr(dk), mem.ptr(dv_), mem.ptr(db) if bias else None,
                                     mem.ptr(dr) if resid else None, Lq, Lk, DV, int(online), mem.ptr(dg), mem.ptr(dbb), mem.ptr(lnout),
                                     mem.ptr(out), mem.stream))
    err = float(np.abs(mem.get(out) - ref.float().numpy()).max())
    assert err <= tol, ("attention", Lq, Lk, DV, bias, resid, online, err)
    if ln:
        lref = F.layer_norm(ref.float().T.contiguous(), (Lq,), torch.from_numpy(gg), torch.from_numpy(bb), 1e-5).T.numpy()
        lerr = float(np.abs(mem.get(lnout) - lref).max())
        assert lerr <= 2 * tol, ("attention+layernorm", Lq, Lk, DV, online, lerr)
    return err


def layernorm(lib, mem, HW, C, seed=0, tol=1e-4):
    g = np.random.default_rng(seed)
    x = (g.standard_normal((HW, C)) * 3 + 1).astype(np.float32)
    gg = g.uniform(0.5, 1.5, HW).astype(np.float32)
    bb = g.standard_normal(HW).astype(np.float32)
    ref = F.layer_norm(torch.from_numpy(x).T.contiguous(), (HW,), torch.from_numpy(gg), torch.from_numpy(bb), 1e-5).T.numpy()
    dx, dg, db, out = mem.put(x), mem.put(gg), mem.put(bb), mem.empty((HW, C))
    lib.check(lib.tdnet_op_layernorm_hw(mem.ptr(dx), HW, C, mem.ptr(dg), mem.ptr(db), mem.ptr(out), mem.stream))
    err = float(np.abs(mem.get(out) - ref).max())
    assert err <= tol, ("layernorm", HW, C, err)
    return err


def ppm(lib, mem, h, w, pid, seed=0, tol=1e-4):
    g = np.random.default_rng(seed)
    c4 = np.abs(g.standard_normal((h, w, 512))).astype(np.float32)
    W4 = (g.standard_normal((4, 128, 512)) * 0.05).astype(np.float32)
    B4 = g.standard_normal((4, 128)).astype(np.float32)
    x = torch.from_numpy(c4).permute(2, 0, 1)[None]
    feats = []
    for j, o in enumerate((1, 2, 3, 6)):
        p = F.relu(F.conv2d(F.adaptive_avg_pool2d(x, o), torch.from_numpy(W4[j])[:, :, None, None], torch.from_numpy(B4[j])))
        feats.append(F.interpolate(p, (h, w), mode="bilinear", align_corners=True))
    ref = torch.cat([x[:, pid * 256:(pid + 1) * 256]] + [f[:, pid * 64:(pid + 1) * 64] for f in feats], 1)[0].permute(1, 2, 0).numpy()
    dc, out = mem.put(c4), mem.empty((h, w, 512))
    lib.check(lib.tdnet_op_ppm(mem.ptr(dc), h, w, W4.ctypes.data, B4.ctypes.data, 2, pid, mem.ptr(out), mem.stream))
    err = float(np.abs(mem.get(out) - ref).max())
    assert err <= tol, ("ppm", h, w, pid, err)
    return err


def upsample(lib, mem, C, h, w, H, W, seed=0, tol=1e-5):
    g = np.random.default_rng(seed)
    x = g.standard_normal((C, h, w)).astype(np.float32)
    ref = F.interpolate(torch.from_numpy(x)[None], (H, W), mode="bilinear", align_corners=True)[0].numpy()
    dx, out = mem.put(x), mem.empty((C, H, W))
    lib.check(lib.tdnet_op_upsample(mem.ptr(dx), C, h, w, H, W, mem.ptr(out), mem.stream))
    err = float(np.abs(mem.get(out) - ref).max())
    assert err <= tol, ("upsample", C, h, w, H, W, err)
    return err
