#!/bin/bash
# round 5, job B: the whole GPU suite on the current library, then the determinism soak on the fixed library
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/jobs/gpu_full_tests.sh r05_full
bash tools/jobs/gpu_r05_soak2.sh ${1:-8000} ${2:-2000} ${3:-3000}
