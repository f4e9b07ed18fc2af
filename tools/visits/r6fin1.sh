#!/bin/bash
# round 6, final visit 1: the whole -m gpu suite and smoke() on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
bash tools/gpu_visit.sh r6fin1 tests smoke
