#!/bin/bash
# round 6, GPU visit s: in-kernel timeline of k_gemm_b3 (s_memtime stamps per wave and K step)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT/gpurun_out/r6s; rm -rf "$R"; mkdir -p "$R"
mkdir -p tools/_build
[ -x tools/_build/gemm_b3_trace ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Iinclude -Itdnet_amd/csrc tools/gemm_b3_trace.hip -o tools/_build/gemm_b3_trace
tools/_build/gemm_b3_trace 2048 512 512 | tee $R/trace_l4.txt | head -80
tools/_build/gemm_b3_trace 2048 512 512 256 | tee $R/trace_l4_one_wg_per_cu.txt | head -40
