#!/bin/bash
# round 5, GPU visit f: does a smaller persistent-GEMM grid (2 workgroups per CU instead of 3) leave the co-resident Winograd transforms more issue slots?
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
R=$GRAFT_REPO_ROOT/gpurun_out/r5f; rm -rf "$R"; mkdir -p "$R"
python -c "import __graft_entry__ as g; g.build()" > $R/build.log 2>&1; tail -n 1 $R/build.log
timeout 900 python tools/ab_opts.py --json $R/ab.jsonl --steps 60 --rounds 3 "" "gemm_persistent=512" "gemm_persistent=640" "gemm_persistent=1024" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp32_gemm_grid.txt
timeout 600 python tools/ab_opts.py --json $R/ab.jsonl --size 769x1537 --steps 60 --rounds 3 "" "gemm_persistent=512" "overlap=0" 2>&1 | grep -v amdgpu.ids | tee $R/ab_fp32_769_gemm_grid.txt
du -sh $R | tail -1
echo "== 769x1537 after the aligned-quad upsample"
timeout 300 python bench.py --steps 60 --quick --size 769x1537 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms', d.get('latency_ms_synced'),'ms synced')"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_769 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --quick --size 769x1537 > $R/prof_769.log 2>&1 )
grep -h "upsample" $(find $R/prof_769 -name "*kernel_stats.csv" | head -1) | cut -c1-140
timeout 600 python -m pytest tests -q -m gpu -k "ops or full_size_digest or vs_oracle_native" 2>&1 | tail -3
find $R -name "*kernel_trace.csv" -delete; find $R -name "*.db" -delete
