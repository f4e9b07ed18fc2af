#!/bin/bash
# round 6, GPU visit ab: the 64-query / eight-wave form of the split attention kernel (k_attention_b3w) against the 32-query form (TDNET_ATTN_B3_NARROW=1)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6ab; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_b3.py -x -q -s -k "split_attention" > $R/test_attn.log 2>&1; grep -E "attention L|passed|failed|Error|assert" $R/test_attn.log | tail -8
cd /tmp && export TMPDIR=/tmp
for v in wide narrow; do
  if [ $v = narrow ]; then export TDNET_ATTN_B3_NARROW=1; else unset TDNET_ATTN_B3_NARROW; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$v -o p -- python $GRAFT_REPO_ROOT/tools/attn_b3_probe.py > $R/probe_$v.log 2>&1
  grep -h "k_attention" $R/prof_$v/*kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-140
done 2>&1 | tee $R/durations.txt
unset TDNET_ATTN_B3_NARROW
cd $GRAFT_REPO_ROOT
grep -h "^Lq" $R/probe_wide.log | tee $R/accuracy_wide.txt
python tools/env_ab.py TDNET_ATTN_B3_NARROW 1024x2048 2>&1 | tail -1 | tee $R/env_ab.txt
python tools/env_ab.py TDNET_ATTN_B3_NARROW 769x1537 2>&1 | tail -1 | tee -a $R/env_ab.txt
find $R -name "*.csv" -size +1M -delete
