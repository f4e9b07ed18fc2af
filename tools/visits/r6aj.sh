#!/bin/bash
# round 6, GPU visit aj: precision-2 tests (incl. the narrow-conv routings) and the model-level GPU tests after the routing change
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/gpu_visit.sh r6aj "tests:b3 or graph or harness or capi" smoke
