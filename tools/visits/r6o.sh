#!/bin/bash
# round 6, GPU visit o: the whole -m gpu suite on the current tree (precision 2 incl. the split attention, product / test libraries, graph tests)
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=2
python -c "import __graft_entry__ as g; g.build()" | tail -2
bash tools/gpu_visit.sh r6o "tests" "smoke"
