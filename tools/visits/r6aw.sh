#!/bin/bash
# round 6, GPU visit aw: precision-2 tests after the deep-K rule; psp101 / td2-psp50 quick lines with parity against the CPU oracle (not --quick: the line's own parity sample)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/gpu_visit.sh r6aw "tests:test_gpu_b3" "quick:--model psp --backbone resnet101 --size 769x1537 --precision bf16x3" "quick:--model td2 --backbone resnet50 --size 769x1537 --precision bf16x3"
