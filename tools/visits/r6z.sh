#!/bin/bash
# round 6, GPU visit z: precision 2 on the 7x7 stem (k_conv_adirect_b3<7, 2>): accuracy against fp64, kernel time, frame A/B with layer1 + stem on the split kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6z; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $R/stem_accuracy.txt
import ctypes, sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "tests")
from tdnet_amd import _capi
lib = _capi.test_lib()
g = np.random.default_rng(0)
for (H, W) in ((1024, 2048), (769, 1537), (131, 259)):
    img = g.standard_normal((3, H, W)).astype(np.float32)
    w = (g.standard_normal((64, 3, 7, 7)) * 0.1).astype(np.float32); b = g.standard_normal(64).astype(np.float32)
    ref = F.max_pool2d(F.relu(F.conv2d(torch.from_numpy(img).double()[None], torch.from_numpy(w).double(), torch.from_numpy(b).double(), 2, 3)), 3, 2, 1)[0].permute(1, 2, 0)
    di = torch.from_numpy(img).cuda()
    res = []
    for kw in ({}, {"precision": 2, "fusion": lib.opts().fusion | 524288}):
        out = torch.full(tuple(ref.shape), 7e7, device="cuda")
        o = lib.opts(**kw)
        outs = []
        for it in range(3):
            lib.check(lib.tdnet_op_stem(di.data_ptr(), H, W, w.ctypes.data, b.ctypes.data, ctypes.byref(o), out.data_ptr(), None))
            outs.append(out.cpu().clone())
        e = (outs[0].double() - ref).abs()
        res.append((e.max().item(), e.pow(2).mean().sqrt().item(), all(torch.equal(outs[0], z) for z in outs[1:]), outs[0]))
    print("stem %dx%d: fp32 max %.2e rms %.2e | split max %.2e rms %.2e | repeatable %s %s | outputs differ %s" % (H, W, res[0][0], res[0][1], res[1][0], res[1][1], res[0][2], res[1][2], not torch.equal(res[0][3], res[1][3])))
PY
cd /tmp && export TMPDIR=/tmp
for p in 0 2; do rocprofv3 --kernel-trace --stats -d $R/stem_prof$p -o s -- python $GRAFT_REPO_ROOT/tools/stem_probe.py $p > $R/stem_prof$p.log 2>&1; grep -E "k_conv_adirect|k_maxpool|k_nchw" $R/stem_prof$p/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120; done
cd $GRAFT_REPO_ROOT
F=$(python -c "from tdnet_amd import _capi; print(_capi.lib().opts().fusion | 524288)" 2>/dev/null | tail -1)
python tools/ab_opts.py --size 1024x2048 --rounds 3 "" "precision=2" "precision=2,fusion=$F" 2>&1 | tail -4 | tee $R/ab_1024.txt
python tools/ab_opts.py --size 769x1537 --rounds 3 "" "precision=2" "precision=2,fusion=$F" 2>&1 | tail -4 | tee $R/ab_769.txt
python tools/ab_opts.py --model td2 --backbone resnet34 --size 720x960 --rounds 2 "" "precision=2" "precision=2,fusion=$F" 2>&1 | tail -4 | tee $R/ab_td2.txt
find $R -name "*.csv" -size +1M -delete
