#!/bin/bash
# round-3 evidence, call 3: the whole GPU suite in one process
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${TAG:-r03_a}
timeout -k 10 1100 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/${T}_gputest.log 2>&1; echo "gpu tests rc=$?"
tail -25 gpurun_out/${T}_gputest.log
