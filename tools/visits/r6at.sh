#!/bin/bash
# round 6, GPU visit at: the new model-level precision-2 gate on a Bottleneck backbone
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/gpu_visit.sh r6at "tests:bottleneck_backbone or forced_on_small"
