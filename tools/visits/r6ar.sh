#!/bin/bash
# round 6, GPU visit ar: precision 2 on every workload of the line (quick lines, parity gated in the run): default vs --precision bf16x3
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/gpu_visit.sh r6ar "quick:--model td2 --backbone resnet18" "quick:--model td2 --backbone resnet18 --precision bf16x3" \
  "quick:--size 769x1537" "quick:--size 769x1537 --precision bf16x3" \
  "quick:--model td2 --backbone resnet50 --size 769x1537" "quick:--model td2 --backbone resnet50 --size 769x1537 --precision bf16x3" \
  "quick:--model psp --backbone resnet101 --size 769x1537" "quick:--model psp --backbone resnet101 --size 769x1537 --precision bf16x3"
