#!/bin/bash
# GPU-box visit: full parity suite + benches + rocprofv3 kernel trace and PMC passes with the default configuration.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=gpurun_out
timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | tail -40 > $R/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1
timeout 600 python bench.py > $R/bench.log 2>&1
timeout 300 python bench.py --winograd 0 --steps 40 --no-cpu-baseline > $R/bench_direct.log 2>&1
timeout 300 python bench.py --model td2 --steps 40 > $R/bench_td2.log 2>&1
timeout 300 python bench.py --size 769x1537 --steps 40 > $R/bench_native.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet50 --size 769x1537 --steps 40 > $R/bench_td2psp50.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet34 --size 720x960 --steps 40 > $R/bench_td2psp34.log 2>&1
timeout 300 python bench.py --model psp --size 769x1537 --steps 30 --cpu-frames 1 > $R/bench_psp101.log 2>&1
timeout 300 python bench.py --clips-per-gpu 3 --steps 30 --no-cpu-baseline > $R/bench_3clips.log 2>&1
timeout 300 python bench.py --mode path-parallel --steps 40 --no-cpu-baseline > $R/bench_pathparallel_n1.log 2>&1
timeout 300 python bench.py --model td2 --backbone resnet34 --size 720x960 --steps 40 --precision fp16 > $R/bench_td2psp34_fp16.log 2>&1
timeout 300 python bench.py --precision fp16 --steps 40 > $R/bench_td4_fp16.log 2>&1
timeout 120 python tools/attn_probe.py > $R/attn_probe.log 2>&1
timeout 120 python tools/wino_probe.py > $R/wino_probe.log 2>&1
timeout 120 python tools/gemm_k_probe.py > $R/gemm_k_probe.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 6 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$R/prof" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/prof.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d "$GRAFT_REPO_ROOT/$R/pmc_sq" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/pmc_sq.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$GRAFT_REPO_ROOT/$R/pmc_fetch" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$GRAFT_REPO_ROOT/$R/pmc_write" -o r1 -- $B > "$GRAFT_REPO_ROOT/$R/pmc_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_prof.py $R > $R/prof_summary.txt 2>&1
find $R -name "*.csv" -size +8M -delete
tail -n 6 $R/gpu_tests.log; tail -n 2 $R/bench.log
