"""Architecture description of the TDNet per-frame hot path (pure data, no torch).

One place that states WHAT the reference builds, so the synthetic-weight generator, the CPU
oracle, the C-ABI weight loader and the tests all agree on names and shapes.

Reference (read-only, /root/reference):
  * backbone layout ......... Testing/model/pspnet/resnet.py:114-202 (_make_layer, dilation / multi-grid rules)
  * td4 wiring .............. Testing/model/pspnet/td4_psp18.py:70-116
  * td2 wiring .............. Testing/model/pspnet/td2_psp50.py:70-89
  * attention module order .. Testing/model/pspnet/td4_psp18.py:145-147,166-168,185-187,204-206
"""
from collections import namedtuple

# kind "basic": conv3x3(cin->cout, stride, dil1) - conv3x3(cout->cout, dil2)               (resnet.py:25-59)
# kind "bottleneck": conv1x1(cin->planes) - conv3x3(planes->planes, stride, dil1) - conv1x1(planes->cout = 4 planes); dil2 unused
#                    (resnet.py:62-111: Bottleneck ignores previous_dilation)
BlockSpec = namedtuple("BlockSpec", "name cin cout stride dil1 dil2 downsample kind planes")

_LAYERS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


def is_bottleneck(backbone):
    return backbone in ("resnet50", "resnet101")


def expansion(backbone):
    return 4 if is_bottleneck(backbone) else 1


def feat_size(n):
    """Spatial size after the three stride-2 stages (stem conv, max-pool, layer2): n -> ((n-1)//2+1) x3."""
    for _ in range(3):
        n = (n - 1) // 2 + 1
    return n


def key_size(n):
    """Size of the stride-4 key/value sub-sampling of a feature axis (transformer.py:26,36)."""
    return (n - 1) // 4 + 1


def backbone_blocks(backbone):
    """BasicBlock list of the dilated, multi-grid ResNet used by TDNet (output stride 8).

    resnet.py:138-149 -> layer1 (64, s1, d1), layer2 (128, s2, d1), layer3 (256, s1, dilation=2),
    layer4 (512, s1, dilation=4, multi_grid -> block dilations 4, 8, 16); conv1 of a block uses `dilation`,
    conv2 uses `previous_dilation` (resnet.py:32-37); the first block of layer3 uses dilation 1 (resnet.py:183-185).
    """
    if backbone not in _LAYERS:
        raise ValueError("backbone must be resnet18/34/50/101; got %r" % (backbone,))
    nb = _LAYERS[backbone]
    bott = is_bottleneck(backbone)
    exp = 4 if bott else 1
    blocks = []
    inpl = 128 if bott else 64                      # deep_base stem ends in 128 channels (resnet.py:117,122-131)
    # (planes, blocks, stride, dilation, multi_grid)
    for li, (planes, n, stride, dil, mg) in enumerate(
            [(64, nb[0], 1, 1, False), (128, nb[1], 2, 1, False), (256, nb[2], 1, 2, False), (512, nb[3], 1, 4, True)], 1):
        for b in range(n):
            first = b == 0
            if mg:
                d1 = (4, 8, 16)[b]
            elif first:
                d1 = 1 if dil in (1, 2) else 2
            else:
                d1 = dil
            ds = first and (stride != 1 or inpl != planes * exp)
            blocks.append(BlockSpec("layer%d.%d" % (li, b), inpl if first else planes * exp, planes * exp,
                                    stride if first else 1, d1, dil, ds, "bottleneck" if bott else "basic", planes))
        inpl = planes * exp
    return blocks


ModelSpec = namedtuple("ModelSpec", "name path_num backbone d_model d_k d_v psp_path_num pids head_mid nclass fifo atn_names")


def model_spec(name, nclass=19, backbone=None):
    """name: 'td4' (td4_psp18.py) or 'td2' (td2_psp50.py with a BasicBlock backbone)."""
    if name == "td4":
        bb = backbone or "resnet18"
        if bb == "resnet101":
            raise ValueError("td4 accepts resnet18 / resnet34 / resnet50 (td4_psp18.py:52)")
        e = expansion(bb)
        # td4_psp18.py:80-83 -> PyramidPooling(512*expansion, path_num=path_num//2, pid=0,1,0,1); :85-88 Encoding(512e, 64, 512e): d_v =
        # d_model = 512 (BasicBlock) or 2048 (resnet50, constructible but never shipped); :112-115 FCNHead(512e, chn_down=4)
        atn = {0: ("atn1_2", "atn1_3", "atn1_4"), 1: ("atn2_3", "atn2_4", "atn2_1"),
               2: ("atn3_4", "atn3_1", "atn3_2"), 3: ("atn4_1", "atn4_2", "atn4_3")}
        return ModelSpec("td4", 4, bb, 512 * e, 64, 512 * e, 2, (0, 1, 0, 1), 512 * e // 4, nclass, 3, atn)
    if name == "td2":
        bb = backbone or "resnet18"
        e = expansion(bb)
        # td2_psp50.py:76-82 -> PyramidPooling(path_num=2, pid=0,1); d_v = 512*exp//4 (128 | 512); head chn_down=2 (:88-89)
        atn = {0: ("atn1",), 1: ("atn2",)}
        return ModelSpec("td2", 2, bb, 512 * e, 64, 128 * e, 2, (0, 1), 128 * e // 2, nclass, 1, atn)
    if name == "psp":
        # pspnet.py:31-70: single-frame PSPNet (the comparison model of test.py:34-38); full pyramid pooling, no attention
        bb = backbone or "resnet101"
        if not is_bottleneck(bb):
            raise ValueError("psp is shipped with resnet101 (pspnet.py:36); only Bottleneck backbones are implemented for it")
        return ModelSpec("psp", 1, bb, 2048, 0, 0, 1, (0,), 512, nclass, 0, {})
    raise ValueError(name)


def state_dict_shapes(spec, h, w):
    """Ordered {key: shape} of the reference state_dict for this model at feature size h x w.

    Mirrors the 728-tensor td4 / td2 checkpoints (names as produced by nn.Module registration order in
    td4_psp18.py:70-116 / td2_psp50.py:70-89). `num_batches_tracked` entries have shape ().
    """
    out = {}

    def bn(prefix, c):
        out[prefix + ".weight"] = (c,)
        out[prefix + ".bias"] = (c,)
        out[prefix + ".running_mean"] = (c,)
        out[prefix + ".running_var"] = (c,)
        out[prefix + ".num_batches_tracked"] = ()

    P = spec.path_num
    bott = is_bottleneck(spec.backbone)
    for p in range(1, P + 1):
        pre = "pretrained%d" % p if spec.name != "psp" else "pretrained"
        if bott:                                     # deep_base stem: resnet.py:122-131
            out[pre + ".conv1.0.weight"] = (64, 3, 3, 3)
            bn(pre + ".conv1.1", 64)
            out[pre + ".conv1.3.weight"] = (64, 64, 3, 3)
            bn(pre + ".conv1.4", 64)
            out[pre + ".conv1.6.weight"] = (128, 64, 3, 3)
            bn(pre + ".bn1", 128)
        else:
            out[pre + ".conv1.weight"] = (64, 3, 7, 7)
            bn(pre + ".bn1", 64)
        for b in backbone_blocks(spec.backbone):
            bp = "%s.%s" % (pre, b.name)
            if b.kind == "basic":
                out[bp + ".conv1.weight"] = (b.cout, b.cin, 3, 3)
                bn(bp + ".bn1", b.cout)
                out[bp + ".conv2.weight"] = (b.cout, b.cout, 3, 3)
                bn(bp + ".bn2", b.cout)
            else:
                out[bp + ".conv1.weight"] = (b.planes, b.cin, 1, 1)
                bn(bp + ".bn1", b.planes)
                out[bp + ".conv2.weight"] = (b.planes, b.planes, 3, 3)
                bn(bp + ".bn2", b.planes)
                out[bp + ".conv3.weight"] = (b.cout, b.planes, 1, 1)
                bn(bp + ".bn3", b.cout)
            if b.downsample:
                out[bp + ".downsample.0.weight"] = (b.cout, b.cin, 1, 1)
                bn(bp + ".downsample.1", b.cout)
        out[pre + ".fc.weight"] = (1000, 512 * expansion(spec.backbone))
        out[pre + ".fc.bias"] = (1000,)
    dm, dk, dv = spec.d_model, spec.d_k, spec.d_v
    if spec.name == "psp":                          # pspnet.py:102-115: PSPHead = PyramidPooling + conv3x3 + BN + ReLU + Dropout + conv1x1
        for j in range(1, 5):
            out["head.conv5.0.conv%d.0.weight" % j] = (dm // 4, dm, 1, 1)
            bn("head.conv5.0.conv%d.1" % j, dm // 4)
        out["head.conv5.1.weight"] = (dm // 4, 2 * dm, 3, 3)
        bn("head.conv5.2", dm // 4)
        out["head.conv5.5.weight"] = (spec.nclass, dm // 4, 1, 1)
        out["head.conv5.5.bias"] = (spec.nclass,)
        return out
    for p in range(1, P + 1):
        for j in range(1, 5):
            out["psp%d.conv%d.0.weight" % (p, j)] = (dm // 4, dm, 1, 1)
            bn("psp%d.conv%d.1" % (p, j), dm // 4)
    for p in range(1, P + 1):
        for br in ("w_qs", "w_ks"):
            out["enc%d.%s.0.conv.weight" % (p, br)] = (dk, dm, 1, 1)
            out["enc%d.%s.0.conv.bias" % (p, br)] = (dk,)
            bn("enc%d.%s.0.bn" % (p, br), dk)
            out["enc%d.%s.1.conv.weight" % (p, br)] = (dk, dk, 1, 1)
            out["enc%d.%s.1.conv.bias" % (p, br)] = (dk,)
        out["enc%d.w_vs.0.conv.weight" % p] = (dv, dm, 1, 1)
        out["enc%d.w_vs.0.conv.bias" % p] = (dv,)
    if spec.name == "td4":
        names = ["atn1_2", "atn1_3", "atn1_4", "atn2_1", "atn2_3", "atn2_4",
                 "atn3_1", "atn3_2", "atn3_4", "atn4_1", "atn4_2", "atn4_3"]
    else:
        names = ["atn1", "atn2"]
    for a in names:
        out[a + ".fc.0.conv.weight"] = (dv, dv, 1, 1)
        out[a + ".fc.0.conv.bias"] = (dv,)
    for p in range(1, P + 1):
        out["layer_norm%d.ln.weight" % p] = (h, w)
        out["layer_norm%d.ln.bias" % p] = (h, w)
    for p in range(1, P + 1):
        out["head%d.conv5.0.weight" % p] = (spec.head_mid, dv, 3, 3)
        bn("head%d.conv5.1" % p, spec.head_mid)
        out["head%d.conv5.4.weight" % p] = (spec.nclass, spec.head_mid, 1, 1)
        out["head%d.conv5.4.bias" % p] = (spec.nclass,)
    return out
