"""CPU numerics experiment behind the choice of Winograd F(4x4,3x3) (test infrastructure; not collected by pytest, run by hand:
`python tests/numerics_winograd.py`).  The oracle's conv2d is swapped, for the wide stride-1 3x3 convs, with an fp32 Winograd
evaluation -- F(2x2), F(4x4) with the interpolation points 0,+-1,+-2 (the kernels' choice) or 0,+-1,+-1/2 -- and the td4 pipeline
is run end to end against an fp64 evaluation of the direct path.  Result on 6 frames at 129x257:
    direct  max|dlogit| 1.20e-05  rms 1.27e-06     F(2x2) 1.24e-05 / 1.46e-06     F(4x4) 2.57e-05 / 2.98e-06     F(4x4, +-1/2) 2.94e-05 / 3.22e-06
i.e. the 16x per-conv rms error of F(4x4) becomes 2x on the logits, against a parity gate of 1e-3."""
import sys; import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import tdnet_ref
from tdnet_amd import arch, weights

def mats(kind):
    if kind == "f2":
        Bt = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
        G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        At = [[1,1,1,0],[0,1,-1,-1]]
        return np.array(Bt, np.float64), np.array(G, np.float64), np.array(At, np.float64), 2
    # general construction from interpolation points (Toom-Cook), m = 4, r = 3, n = 6 points incl. infinity
    pts = {"f4": [0, 1, -1, 2, -2], "f4h": [0, 1, -1, 0.5, -0.5]}[kind]
    m, r = 4, 3
    n = m + r - 1
    # Vandermonde-based: A^T (m x n), G (n x r), B^T (n x n) such that Y = A^T [(G g) * (B^T d)]
    a = np.array(pts, np.float64)
    # polynomial method: evaluate at points; infinity is the last row
    At = np.zeros((m, n)); G = np.zeros((n, r)); 
    for i, p in enumerate(a):
        for j in range(m): At[j, i] = p ** j
        for j in range(r): G[i, j] = p ** j
    At[m - 1, n - 1] = 1.0; G[n - 1, r - 1] = 1.0
    # scale rows of G by 1/prod(p_i - p_k)
    for i, p in enumerate(a):
        den = np.prod([p - q for k, q in enumerate(a) if k != i])
        G[i] /= den
    # B^T from the Lagrange basis: B^T = rows such that sum_k B^T[i,k] x^k = prod_{k != i}(x - p_k) ... solve numerically:
    # requirement: for all d (len n), g (len r): A^T[(G g)*(B^T d)] = conv(d, g) valid part. Solve linear system for B^T.
    # unknown B^T (n*n). Build equations with basis vectors.
    rows = []; rhs = []
    for gi in range(r):
        g = np.zeros(r); g[gi] = 1
        Gg = G @ g
        for di in range(n):
            d = np.zeros(n); d[di] = 1
            y = np.array([sum(d[o + t] * g[t] for t in range(r)) for o in range(m)])
            # y = At @ (Gg * (Bt @ d)) = At @ diag(Gg) @ Bt[:, di]
            M = At @ np.diag(Gg)
            for o in range(m):
                row = np.zeros(n * n)
                for i in range(n): row[i * n + di] = M[o, i]
                rows.append(row); rhs.append(y[o])
    sol, res, rk, sv = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)
    Bt = sol.reshape(n, n)
    return Bt, G, At, 4

def check(kind):
    Bt, G, At, m = mats(kind)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(Bt.shape[0]); g = rng.standard_normal(3)
    y = At @ ((G @ g) * (Bt @ d))
    ref = np.array([sum(d[o + t] * g[t] for t in range(3)) for o in range(m)])
    return np.abs(y - ref).max()

class Wino:
    def __init__(self, kind, dtype=torch.float32):
        Bt, G, At, m = mats(kind)
        self.m = m; self.n = Bt.shape[0]
        self.Bt = torch.tensor(Bt, dtype=dtype); self.At = torch.tensor(At, dtype=dtype)
        self.G = torch.tensor(G, dtype=torch.float64); self.dtype = dtype
    def conv(self, x, w, dil):
        # x [1,C,H,W], w [O,C,3,3]; dilation via sub-grid decomposition; 'same' padding
        _, C, H, W = x.shape; O = w.shape[0]
        U = torch.einsum("ij,ocjk,lk->iloc", self.G, w.double(), self.G).to(self.dtype)      # [n,n,O,C], fp64 then rounded (host side)
        out = torch.zeros(1, O, H, W, dtype=self.dtype)
        m, n = self.m, self.n
        for py in range(dil):
            for px in range(dil):
                sub = x[:, :, py::dil, px::dil]
                h, wd = sub.shape[2], sub.shape[3]
                ty, tx = (h + m - 1) // m, (wd + m - 1) // m
                padded = F.pad(sub, (1, tx * m + 1 - wd, 1, ty * m + 1 - h))
                patches = padded.unfold(2, n, m).unfold(3, n, m)                              # [1,C,ty,tx,n,n]
                V = torch.einsum("ij,bcyxjk,lk->ilyxc", self.Bt, patches, self.Bt)           # [n,n,ty,tx,C]
                Mm = torch.einsum("ilyxc,iloc->ilyxo", V, U)
                Y = torch.einsum("ai,ilyxo,bl->oyaxb", self.At, Mm, self.At).reshape(1, O, ty * m, tx * m)
                out[:, :, py::dil, px::dil] = Y[:, :, :h, :wd]
        return out

def run(kind, H=129, W=257, T=6, dtype=torch.float32, init="calibrated", cmin=256):
    spec = arch.model_spec("td4", 19, "resnet18")
    h, w = arch.feat_size(H), arch.feat_size(W)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in weights.synth_state_dict(spec, h, w, 0, init).items()}
    if dtype == torch.float64:
        sd = {k: v.double() for k, v in sd.items()}
    ref = tdnet_ref.TDNetRef(spec, sd)
    orig = tdnet_ref.OPS.conv2d
    wn = Wino(kind, dtype) if kind != "direct" else None
    def patched(x, wt, bias=None, stride=1, padding=0, dilation=1, groups=1):
        dl = dilation if isinstance(dilation, int) else dilation[0]
        st = stride if isinstance(stride, int) else stride[0]
        if wn is not None and wt.shape[2] == 3 and st == 1 and wt.shape[1] >= cmin and wt.shape[0] >= 128:
            y = wn.conv(x, wt, dl)
            return y if bias is None else y + bias.view(1, -1, 1, 1)
        return orig(x, wt, bias, stride, padding, dilation, groups)
    class Patched(tdnet_ref.TorchOps):
        conv2d = staticmethod(patched)
    prev = tdnet_ref.set_ops(Patched)
    outs = []
    try:
        for t, x in enumerate(weights.synth_video(H, W, T, seed=1)):
            xt = torch.from_numpy(x)
            if dtype == torch.float64: xt = xt.double()
            outs.append(ref.forward(xt, t % 4).double().numpy())
    finally:
        tdnet_ref.set_ops(prev)
    return outs

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "reference-init":
    # SURVEY 8d's original init (no calibration): what matters is the error RELATIVE to the logits' scale
    torch.set_num_threads(8)
    truth = run("direct", dtype=torch.float64, init="reference", T=5)
    scale = float(np.sqrt(np.mean([np.mean(a ** 2) for a in truth])))
    print("reference init: logits rms %.3e, max %.3e" % (scale, max(np.abs(a).max() for a in truth)))
    for kind in ("direct", "f2", "f4"):
        o = run(kind, init="reference", T=5, cmin=128)
        err = max(np.abs(a - b).max() for a, b in zip(o, truth))
        rms = np.sqrt(np.mean([np.mean((a - b) ** 2) for a, b in zip(o, truth)]))
        flips = sum(int((a[0].argmax(0) != b[0].argmax(0)).sum()) for a, b in zip(o, truth))
        print("%-7s vs fp64 truth: max|dlogit| %.3e (%.2e of rms)  rms err %.3e (%.2e of rms)  label flips %d" % (kind, err, err / scale, rms, rms / scale, flips), flush=True)
    sys.exit(0)
if __name__ == "__main__":
    torch.set_num_threads(32)
    for k in ("f2", "f4", "f4h"):
        print(k, "identity check", check(k))
    truth = run("direct", dtype=torch.float64)
    for kind in ("direct", "f2", "f4h", "f4"):
        o = run(kind)
        err = max(np.abs(a - b).max() for a, b in zip(o, truth))
        rms = np.sqrt(np.mean([np.mean((a - b) ** 2) for a, b in zip(o, truth)]))
        flips = sum(int((a[0].argmax(0) != b[0].argmax(0)).sum()) for a, b in zip(o, truth))
        print("%-7s vs fp64 truth: max|dlogit| %.3e  rms %.3e  label flips %d / %d" % (kind, err, rms, flips, len(o) * o[0].shape[2] * o[0].shape[3]), flush=True)
