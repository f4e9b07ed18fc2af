#!/bin/bash
# round 6, GPU visit u: k_gemm_b3 with A loaded straight from global memory into registers (TDNET_B3_AG=1, experiment) against the shipped form
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6u; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1
TDNET_B3_AG=1 timeout 600 python -m pytest tests/test_gpu_b3.py -x -q -k "operators or gate_1024 or forced" > $R/test_ag.log 2>&1; tail -3 $R/test_ag.log
for i in 1 2 3; do
  python tools/b3_one.py 5 512 512 4 20 >> $R/one.txt 2>&1
  TDNET_B3_AG=1 python tools/b3_one.py 5 512 512 4 20 >> $R/one_ag.txt 2>&1
done
cat $R/one.txt $R/one_ag.txt
for c in "256 256 2" "128 128 1"; do set -- $c; python tools/b3_one.py 5 $1 $2 $3 20; TDNET_B3_AG=1 python tools/b3_one.py 5 $1 $2 $3 20; done 2>&1 | tee $R/one_small.txt
tools/_build/gemm_b3_trace 2048 512 512 0 0 > $R/trace.txt; head -6 $R/trace.txt
TDNET_B3_AG=1 tools/_build/gemm_b3_trace 2048 512 512 0 0 > $R/trace_ag.txt; head -6 $R/trace_ag.txt
python tools/env_ab.py TDNET_B3_AG 1024x2048 2>&1 | tail -1 | tee $R/env_ab.txt
python tools/env_ab.py TDNET_B3_AG 769x1537 2>&1 | tail -1 | tee -a $R/env_ab.txt
