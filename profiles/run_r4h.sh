#!/bin/bash
# round 4: per-kernel timeline of one iteration with the deep chains (forward + backward)
set -o pipefail
O=gpurun_out/r4h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEP="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --no-extra-modes"
export LSHM_WGRAD_GROUP=2
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/raw_step --output-format csv -- python3 bench.py $STEP > $O/step.json 2> $O/step.err || exit 1
python3 profiles/step_trace.py $O/raw_step > $O/step_timeline.txt
head -1 $O/step_timeline.txt
rm -rf $O/raw_step
