#!/bin/bash
# GPU-box visit: bench lines of every workload (with CPU baseline + parity where stated) for the record.
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out/${1:-sweep}
rm -rf $R; mkdir -p $R
Q="--no-pmc --no-direct-line"
timeout 600 python bench.py --model td2 --steps 40 $Q > $R/02_td2psp18_c2.log 2>&1
timeout 300 python bench.py --size 769x1537 --steps 40 $Q > $R/03_td4_native.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet50 --size 769x1537 --steps 40 $Q > $R/04_td2psp50.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet34 --size 720x960 --steps 40 $Q > $R/05_td2psp34.log 2>&1
timeout 300 python bench.py --model psp --size 769x1537 --steps 30 --cpu-frames 1 $Q > $R/06_psp101.log 2>&1
timeout 300 python bench.py --winograd 1 --steps 40 --no-cpu-baseline $Q > $R/08_td4_f2.log 2>&1
timeout 300 python bench.py --clips-per-gpu 2 --steps 30 --no-cpu-baseline $Q > $R/09_td4_2clips.log 2>&1
timeout 300 python bench.py --mode path-parallel --steps 40 --no-cpu-baseline $Q > $R/10_td4_pathparallel_n1.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet34 --size 720x960 --steps 40 --precision fp16 $Q > $R/11_td2psp34_fp16.log 2>&1
timeout 300 python bench.py --precision fp16 --steps 40 $Q > $R/12_td4_fp16.log 2>&1
timeout 300 python bench.py --model td4 --backbone resnet34 --steps 40 --no-cpu-baseline $Q > $R/13_td4psp34.log 2>&1
timeout 300 python bench.py --model td4 --backbone resnet50 --size 769x1537 --steps 30 --cpu-frames 1 $Q > $R/14_td4psp50.log 2>&1
for f in $R/*.log; do tail -1 $f; done > $R/lines.jsonl
python - <<PY
import json
for l in open("$R/lines.jsonl"):
    try:
        d = json.loads(l)
        p = d.get("parity", {})
        print("%-70s %9.2f fps %7.3f ms  parity %s  cpu %s" % (d["metric"][12:] + (" " + d["dtype"][:3] if d["dtype"] != "f32" else "") + (" | " + d["config"]["parallelism"][:14]), d["value"], d["ms_per_step"],
              (p.get("max_abs_dlogit"), p.get("label_mismatches"), p.get("flips_outside_tie_band"), p.get("miou_vs_cpu")) if p else "-", d.get("cpu_baseline", {}).get("value", "-")))
    except Exception as e:
        print("BAD LINE", l[:200])
PY
