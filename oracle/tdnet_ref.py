"""CPU ORACLE (test infrastructure, NOT product code) -- fp32 restatement of TDNet's per-frame hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  The product
package (tdnet_amd/) never does: it fails loudly when the HIP library is missing.

What it is: a functional re-statement, written from scratch, of the reference's Testing/ model graph on
torch CPU tensors.  The reference's arithmetic lives in PyTorch, so the graph is written over a
small set of L0 operators -- conv2d / max-pool / adaptive avg-pool / interpolate / bmm / softmax / layer_norm --
with two implementations: TorchOps (default: PyTorch's own CPU kernels, what the reference runs on; fast enough
for 1024x2048) and oracle/c_ops.COps (oracle/ops_c.c: the same operators restated in plain C, so that no
PyTorch kernel is left on the checking side; set_ops() swaps them).
State is an explicit FIFO, weights are a flat {name: tensor} dict with the reference's state_dict keys.

Pinning: tools/make_golden.py imports the real reference from /root/reference (this container only),
loads the same synthetic weights, and stores per-stage outputs under tests/golden/; tests/test_oracle.py
checks this file against those vectors.  Parity is pinned by those goldens (the reference has no tests).

Every function cites the reference lines it follows (paths relative to /root/reference/Testing/model/pspnet).
"""
import math
import os
from collections import namedtuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5

# ---- architecture facts, restated HERE from the reference -- the checker does not take its graph description from the product
# package (tdnet_amd/arch.py states the same facts for the HIP side; tests/test_oracle.py cross-checks the two statements) ----------
RefBlock = namedtuple("RefBlock", "name kind stride dil1 dil2 downsample")

_BLOCK_COUNTS = {"resnet18": ("basic", (2, 2, 2, 2)), "resnet34": ("basic", (3, 4, 6, 3)),          # resnet.py:218-256
                 "resnet50": ("bottleneck", (3, 4, 6, 3)), "resnet101": ("bottleneck", (3, 4, 23, 3))}


def ref_backbone_blocks(backbone):
    """Block list of ResNet(dilated=True, multi_grid=True) as resnet.py:138-149 builds it through _make_layer (:172-202):
    layer1 (64), layer2 (128, stride 2), layer3 (256, stride 1, dilation 2), layer4 (512, stride 1, dilation 4, multi_grid).
    `dil1` is the block's `dilation` (conv1 of a BasicBlock / the 3x3 of a Bottleneck), `dil2` its `previous_dilation` (conv2 of a
    BasicBlock, resnet.py:32-37; unused by Bottleneck)."""
    kind, counts = _BLOCK_COUNTS[backbone]
    expansion = 4 if kind == "bottleneck" else 1
    inplanes = 128 if kind == "bottleneck" else 64                  # deep_base for the Bottleneck nets (td2_psp50.py:63-66, resnet.py:118)
    out = []
    for li, (planes, nblocks, stride, dilation, multi_grid) in enumerate(
            ((64, counts[0], 1, 1, False), (128, counts[1], 2, 1, False), (256, counts[2], 1, 2, False), (512, counts[3], 1, 4, True)), 1):
        downsample = stride != 1 or inplanes != planes * expansion      # resnet.py:173-178
        if multi_grid:                                                  # :182-184
            first = 4
        elif dilation in (1, 2):                                        # :185-187
            first = 1
        elif dilation == 4:                                             # :188-190
            first = 2
        else:
            raise RuntimeError("=> unknown dilation size: {}".format(dilation))
        out.append(RefBlock("layer%d.0" % li, kind, stride, first, dilation, downsample))
        inplanes = planes * expansion
        for i in range(1, nblocks):                                     # :195-200
            out.append(RefBlock("layer%d.%d" % (li, i), kind, 1, (4, 8, 16)[i] if multi_grid else dilation, dilation, False))
    return out


# Attention modules in APPLICATION order per sub-network (0-based path index): forward_path1..4 of td4_psp18.py
#   path1 :145-147   atn1_2(K[0],V[0],Q[1]) -> atn1_3 -> atn1_4        path2 :166-168   atn2_3 -> atn2_4 -> atn2_1
#   path3 :185-187   atn3_4 -> atn3_1 -> atn3_2                        path4 :204-206   atn4_1 -> atn4_2 -> atn4_3
# and td2_psp50.py:120 (atn1) / :138 (atn2).
REF_ATN_ORDER = {"td4": (("atn1_2", "atn1_3", "atn1_4"), ("atn2_3", "atn2_4", "atn2_1"),
                         ("atn3_4", "atn3_1", "atn3_2"), ("atn4_1", "atn4_2", "atn4_3")),
                 "td2": (("atn1",), ("atn2",))}
# PyramidPooling(path_num, pid) per sub-network: td4_psp18.py:80-83 (path_num//2 = 2; pid 0,1,0,1), td2_psp50.py:76-77 (2; pid 0,1)
REF_PSP = {"td4": (2, (0, 1, 0, 1)), "td2": (2, (0, 1))}
# FIFO depth: buffer_contral pops when len > 3 (td4_psp18.py:130) / > 1 (td2_psp50.py:105)
REF_FIFO = {"td4": 3, "td2": 1}


class TorchOps:
    """Default L0 ops: PyTorch's own CPU kernels -- the arithmetic the reference itself runs on."""
    conv2d = staticmethod(F.conv2d)
    relu = staticmethod(F.relu)
    leaky_relu = staticmethod(F.leaky_relu)
    max_pool2d = staticmethod(F.max_pool2d)
    adaptive_avg_pool2d = staticmethod(F.adaptive_avg_pool2d)
    interpolate = staticmethod(F.interpolate)
    layer_norm = staticmethod(F.layer_norm)
    bmm = staticmethod(torch.bmm)
    matmul = staticmethod(torch.matmul)
    softmax = staticmethod(torch.softmax)


OPS = TorchOps


def set_ops(ops):
    """Swap the L0 ops under the graph: TorchOps (default) or oracle.c_ops.COps -- the plain-C restatement of the same ops
    (oracle/ops_c.c), which makes the oracle independent of PyTorch's kernels.  Returns the previous backend."""
    global OPS
    prev, OPS = OPS, ops
    return prev


def bn_eval(x, sd, pre, leaky=False):
    """Eval-mode BatchNorm2d followed by identity or LeakyReLU(0.01): td4_psp18.py:11-24."""
    y = (x - sd[pre + ".running_mean"][None, :, None, None]) / torch.sqrt(sd[pre + ".running_var"][None, :, None, None] + BN_EPS)
    y = y * sd[pre + ".weight"][None, :, None, None] + sd[pre + ".bias"][None, :, None, None]
    return OPS.leaky_relu(y, 0.01) if leaky else y


def basic_block(x, sd, pre, b):
    """resnet.py:25-59: relu(bn1(conv1 x)) -> bn2(conv2 .) -> + (downsample x | x) -> relu."""
    out = OPS.conv2d(x, sd[pre + ".conv1.weight"], None, b.stride, b.dil1, b.dil1)
    out = OPS.relu(bn_eval(out, sd, pre + ".bn1"))
    out = OPS.conv2d(out, sd[pre + ".conv2.weight"], None, 1, b.dil2, b.dil2)
    out = bn_eval(out, sd, pre + ".bn2")
    res = x
    if b.downsample:
        res = bn_eval(OPS.conv2d(x, sd[pre + ".downsample.0.weight"], None, b.stride), sd, pre + ".downsample.1")
    return OPS.relu(out + res)


def bottleneck_block(x, sd, pre, b):
    """resnet.py:62-111: 1x1 -> BN -> ReLU -> 3x3(stride, dilation) -> BN -> ReLU -> 1x1 (x4) -> BN -> + residual -> ReLU."""
    out = OPS.relu(bn_eval(OPS.conv2d(x, sd[pre + ".conv1.weight"]), sd, pre + ".bn1"))
    out = OPS.relu(bn_eval(OPS.conv2d(out, sd[pre + ".conv2.weight"], None, b.stride, b.dil1, b.dil1), sd, pre + ".bn2"))
    out = bn_eval(OPS.conv2d(out, sd[pre + ".conv3.weight"]), sd, pre + ".bn3")
    res = x
    if b.downsample:
        res = bn_eval(OPS.conv2d(x, sd[pre + ".downsample.0.weight"], None, b.stride), sd, pre + ".downsample.1")
    return OPS.relu(out + res)


def backbone(x, sd, pre, blocks):
    """resnet.py:204-215: stem (7x7 s2 p3, or the deep_base 3-conv stem :122-131), BN, ReLU, max-pool 3x3 s2 p1, layer1..4 -> c4."""
    if pre + ".conv1.0.weight" in sd:               # deep_base (ResNet-50)
        x = OPS.relu(bn_eval(OPS.conv2d(x, sd[pre + ".conv1.0.weight"], None, 2, 1), sd, pre + ".conv1.1"))
        x = OPS.relu(bn_eval(OPS.conv2d(x, sd[pre + ".conv1.3.weight"], None, 1, 1), sd, pre + ".conv1.4"))
        x = OPS.conv2d(x, sd[pre + ".conv1.6.weight"], None, 1, 1)
    else:
        x = OPS.conv2d(x, sd[pre + ".conv1.weight"], None, 2, 3)
    x = OPS.relu(bn_eval(x, sd, pre + ".bn1"))
    x = OPS.max_pool2d(x, 3, 2, 1)
    for b in blocks:
        x = (bottleneck_block if b.kind == "bottleneck" else basic_block)(x, sd, "%s.%s" % (pre, b.name), b)
    return x


def pyramid_pooling(c4, sd, pre, path_num, pid):
    """td4_psp18.py:271-284: 4 adaptive pools (1,2,3,6) -> 1x1 conv+BN+ReLU -> bilinear(align_corners) -> slice+concat."""
    n, c, h, w = c4.shape
    feats = []
    for j, o in enumerate((1, 2, 3, 6), 1):
        p = OPS.adaptive_avg_pool2d(c4, o)
        p = OPS.relu(bn_eval(OPS.conv2d(p, sd["%s.conv%d.0.weight" % (pre, j)]), sd, "%s.conv%d.1" % (pre, j)))
        feats.append(OPS.interpolate(p, (h, w), mode="bilinear", align_corners=True))
    cs = c // path_num
    fs = c // (path_num * 4)
    parts = [c4[:, pid * cs:(pid + 1) * cs]] + [f[:, pid * fs:(pid + 1) * fs] for f in feats]
    return torch.cat(parts, 1)


def _conv_bias(x, sd, pre):
    """ConvBNReLU with norm_layer=None is just a biased 1x1 conv: transformer.py:142-161."""
    return OPS.conv2d(x, sd[pre + ".conv.weight"], sd[pre + ".conv.bias"])


def _qk_branch(x, sd, pre):
    """w_qs / w_ks: 1x1 conv(+bias) -> BN -> LeakyReLU(0.01) -> 1x1 conv(+bias): transformer.py:18-22."""
    y = bn_eval(_conv_bias(x, sd, pre + ".0"), sd, pre + ".0.bn", leaky=True)
    return _conv_bias(y, sd, pre + ".1")


def _flat(x):
    """[n,c,h,w] -> [n, h*w, c] (row-major positions): transformer.py:43-49."""
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).contiguous().view(n, h * w, c)


def encoding(z, sd, pre, pre_flag):
    """transformer.py:28-56.  pre=False -> (q [n,Lq,dk], v [n,dv,h,w]); pre=True -> stride-4 subsample, (q_,k_,v_) flat."""
    if pre_flag:
        zs = z[:, :, ::4, ::4]                      # MaxPool2d(kernel 1, stride 4): transformer.py:26,36
        k_ = _flat(_qk_branch(zs, sd, pre + ".w_ks"))
        v_ = _flat(_conv_bias(zs, sd, pre + ".w_vs.0"))
        q_ = _flat(_qk_branch(zs, sd, pre + ".w_qs"))
        return q_, k_, v_
    v = _conv_bias(z, sd, pre + ".w_vs.0")
    q = _flat(_qk_branch(z, sd, pre + ".w_qs"))
    return q, v


def attention(k_src, v_src, q_tgr, sd, pre, fea_size=None):
    """transformer.py:71-92 + :126-139: softmax(q k^T / 8) v, then per-position fc (1x1 conv with bias)."""
    dk = q_tgr.shape[-1]
    attn = OPS.bmm(q_tgr, k_src.transpose(1, 2)) / math.pow(dk, 0.5)
    attn = OPS.softmax(attn, dim=2)
    out = OPS.bmm(attn, v_src)                                    # [n, Lq, dv]
    w = sd[pre + ".fc.0.conv.weight"][:, :, 0, 0]
    out = OPS.matmul(out, w.t()) + sd[pre + ".fc.0.conv.bias"]
    if fea_size is not None:
        n, _, h, wd = fea_size
        out = out.permute(0, 2, 1).contiguous().view(n, -1, h, wd)
    return out


def layer_norm_hw(x, sd, pre):
    """td4_psp18.py:306-312: nn.LayerNorm([h,w]) -- each channel plane normalised, affine [h,w] shared over channels."""
    return OPS.layer_norm(x, x.shape[-2:], sd[pre + ".ln.weight"], sd[pre + ".ln.bias"], 1e-5)


def fcn_head(x, sd, pre):
    """td4_psp18.py:287-302: conv3x3 p1 (no bias) -> BN -> ReLU -> Dropout2d(eval=id) -> conv1x1 (+bias)."""
    y = OPS.relu(bn_eval(OPS.conv2d(x, sd[pre + ".conv5.0.weight"], None, 1, 1), sd, pre + ".conv5.1"))
    return OPS.conv2d(y, sd[pre + ".conv5.4.weight"], sd[pre + ".conv5.4.bias"])


class TDNetRef:
    """Stateful per-frame forward: td4_psp18.py:123-229 / td2_psp50.py:98-155 (FIFO depth P-1, warm-up branch)."""

    def __init__(self, spec, state_dict):
        """spec: anything with .name ("td4" | "td2") and .backbone ("resnet18" ...); everything else the graph needs is restated
        above from the reference."""
        self.spec = spec
        self.name, self.path_num = spec.name, {"td4": 4, "td2": 2}[spec.name]
        self.fifo = REF_FIFO[spec.name]
        self.psp_path_num, self.pids = REF_PSP[spec.name]
        self.atn_order = REF_ATN_ORDER[spec.name]
        self.sd = {k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in state_dict.items()}
        self.blocks = ref_backbone_blocks(spec.backbone)
        self.Q, self.K, self.V = [], [], []
        self.trace = None            # optional dict filled with stage outputs of the last frame

    def reset(self):
        self.Q, self.K, self.V = [], [], []

    def _t(self, name, val):
        if self.trace is not None:
            self.trace[name] = val

    def forward_lowres(self, img, pos_id):
        sd = self.sd
        p = pos_id + 1
        c4 = backbone(img, sd, "pretrained%d" % p, self.blocks)
        z = pyramid_pooling(c4, sd, "psp%d" % p, self.psp_path_num, self.pids[pos_id])
        q_cur, v_cur = encoding(z, sd, "enc%d" % p, False)
        self._t("c4", c4); self._t("z", z); self._t("q_cur", q_cur); self._t("v_cur", v_cur)
        if len(self.Q) < self.fifo:                                 # warm-up: td4_psp18.py:142-143
            feat = v_cur
        else:
            names = self.atn_order[pos_id]
            if self.name == "td4":                                   # td4_psp18.py:145-151
                v2 = attention(self.K[0], self.V[0], self.Q[1], sd, names[0])
                v3 = attention(self.K[1], v2 + self.V[1], self.Q[2], sd, names[1])
                v4 = attention(self.K[2], v3 + self.V[2], q_cur, sd, names[2], fea_size=z.shape)
                self._t("v2", v2); self._t("v3", v3); self._t("v4", v4)
                feat = v4 + v_cur
            else:                                                    # td2_psp50.py:120-122
                v1 = attention(self.K[0], self.V[0], q_cur, sd, names[0], fea_size=z.shape)
                self._t("v4", v1)
                feat = v1 + v_cur
        ln = layer_norm_hw(feat, sd, "layer_norm%d" % p)
        out = fcn_head(ln, sd, "head%d" % p)
        self._t("feat", feat); self._t("ln", ln); self._t("lowres", out)
        q_, k_, v_ = encoding(z, sd, "enc%d" % p, True)              # td4_psp18.py:153-154
        self._t("cache_q", q_); self._t("cache_k", k_); self._t("cache_v", v_)
        self.Q.append(q_); self.K.append(k_); self.V.append(v_)
        if len(self.Q) > self.fifo:                                 # buffer_contral: td4_psp18.py:123-134
            self.Q.pop(0); self.K.pop(0); self.V.pop(0)
        return out

    @torch.no_grad()
    def forward(self, img, pos_id=0):
        """td4_psp18.py:216-229: path dispatch, then bilinear align_corners upsample to the input size."""
        h, w = img.shape[-2:]
        out = self.forward_lowres(img, pos_id)
        return OPS.interpolate(out, (h, w), mode="bilinear", align_corners=True)


class PSPNetRef:
    """Stateless single-frame PSPNet (the comparison model `--model psp101`, test.py:34-38): pspnet.py:73-89 forward,
    PSPHead pspnet.py:102-115 = full pyramid pooling (:118-157, no slicing) -> conv3x3 -> BN -> ReLU -> conv1x1."""

    def __init__(self, spec, state_dict):
        self.spec = spec
        self.sd = {k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in state_dict.items()}
        self.blocks = ref_backbone_blocks(spec.backbone)
        self.trace = None

    def reset(self):
        pass

    @torch.no_grad()
    def forward(self, img, pos_id=None):
        sd = self.sd
        h, w = img.shape[-2:]
        c4 = backbone(img[-1:], sd, "pretrained", self.blocks)
        z = pyramid_pooling(c4, sd, "head.conv5.0", 1, 0)             # path_num 1, pid 0 = every channel (pspnet.py:157)
        y = OPS.relu(bn_eval(OPS.conv2d(z, sd["head.conv5.1.weight"], None, 1, 1), sd, "head.conv5.2"))
        low = OPS.conv2d(y, sd["head.conv5.5.weight"], sd["head.conv5.5.bias"])
        if self.trace is not None:
            self.trace.update(c4=c4, z=z, lowres=low)
        return OPS.interpolate(low, (h, w), mode="bilinear", align_corners=True)


def tune_threads(candidates=(16, 32, 48, 64, 96, 128)):
    """Pick the torch-CPU thread count that runs the path's dominant conv (layer4: 512 -> 512, 3x3, dilation 4, at 1/8 of
    1024x2048) fastest: oneDNN collapses when it is given every SMT thread of a 256-thread host (measured 42 s/frame vs ~2.5 s),
    and the best count moves between 32 and 128 from box to box.  Best of three runs per candidate (a single run is noisy enough to
    pick a count that is 2x slower on the whole frame).  Returns the count set."""
    import time
    cores = os.cpu_count() or 1
    cands = sorted({min(c, cores) for c in candidates} | ({8} if cores <= 16 else set()))
    x, w = torch.randn(1, 512, 128, 256), torch.randn(512, 512, 3, 3)
    best, best_t = cands[0], None
    for n in cands:
        torch.set_num_threads(n)
        F.conv2d(x, w, None, 1, 4, 4)
        dt = None
        for _ in range(3):
            t0 = time.perf_counter()
            F.conv2d(x, w, None, 1, 4, 4)
            d = time.perf_counter() - t0
            dt = d if dt is None or d < dt else dt
        if best_t is None or dt < best_t:
            best, best_t = n, dt
    torch.set_num_threads(best)
    return best


def confusion_miou(pred, ref, n_class):
    """mIoU of `pred` against `ref` labels via the confusion-matrix formula of Training/ptsemseg/metrics.py:12-35."""
    import numpy as np
    pred = np.asarray(pred).reshape(-1); ref = np.asarray(ref).reshape(-1)
    hist = np.bincount(n_class * ref.astype(np.int64) + pred.astype(np.int64), minlength=n_class ** 2).reshape(n_class, n_class)
    with np.errstate(divide="ignore", invalid="ignore"):
        iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))
    return float(np.nanmean(iu)), hist
