#!/bin/bash
# round 6, GPU visit ag: precision 2 on the reference's Bottleneck configurations at its native size (td2-psp50, psp101 @769x1537) and kernel stats of td2-psp50
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r6ag; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
python tools/ab_opts.py --model td2 --backbone resnet50 --size 769x1537 --steps 30 --rounds 2 "" "precision=2" 2>&1 | tail -3 | tee $R/ab_td2psp50.txt
python tools/ab_opts.py --model psp --backbone resnet101 --size 769x1537 --steps 20 --rounds 2 "" "precision=2" 2>&1 | tail -3 | tee $R/ab_psp101.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_td2psp50_p2 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --model td2 --backbone resnet50 --size 769x1537 --precision bf16x3 > $R/prof.log 2>&1
head -16 $R/prof_td2psp50_p2/*kernel_stats.csv | cut -c1-150
find $R -name "*kernel_trace.csv" -delete
