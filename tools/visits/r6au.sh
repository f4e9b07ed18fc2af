#!/bin/bash
# round 6, GPU visit au: kernel stats of psp101 @769x1537 with precision 2
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/gpu_visit.sh r6au "prof:psp101_b3:--model psp --backbone resnet101 --size 769x1537 --precision bf16x3"
