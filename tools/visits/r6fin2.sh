#!/bin/bash
# round 6, final visit 2: the driver's line (default flags), the same with --precision bf16x3, rocprofv3 kernel stats + timelines of both, MFMA-busy counters of both
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/gpu_visit.sh r6fin2 "bench:--steps 20 --warmup 5" "quick:--precision bf16x3" "prof:fp32" "prof:b3:--precision bf16x3" "cmd:bash tools/gpu_pmc.sh r6fin2_pmc_fp32" "cmd:bash tools/gpu_pmc.sh r6fin2_pmc_b3 --precision bf16x3"
