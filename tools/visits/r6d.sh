#!/bin/bash
# round 6, GPU visit d: epilogue-store probes of k_gemm_b3 (probe build)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2 TDNET_EXTRA_CXXFLAGS=-DTD_B3_PROBE
R=$GRAFT_REPO_ROOT/gpurun_out/r6d; rm -rf "$R"; mkdir -p "$R"
( time python -c "import __graft_entry__ as g; g.build()" ) > $R/build.log 2>&1; tail -n 5 $R/build.log
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/prof_b3" -o r1 -- python $GRAFT_REPO_ROOT/tools/b3_probe.py > "$R/b3_probe.txt" 2>&1 )
grep -v amdgpu.ids $R/b3_probe.txt
cp $(find $R/prof_b3 -name "*kernel_stats.csv" | head -1) $R/kernel_stats_b3_probe.csv 2>/dev/null; grep "k_gemm_b3\|k_gemm_dma\|wino" $R/kernel_stats_b3_probe.csv | cut -c1-160
rm -rf $R/prof_b3
