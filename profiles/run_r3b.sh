#!/bin/bash
O=gpurun_out/r3b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
Q="--no-extra-modes --no-roofline --no-cpu-baseline --no-lbfgs --no-rica --no-reuse-mode --steps 40 --warmup 5"
for hq in default 8; do
  for mode in "--sequential-forwards" ""; do
    if [ $hq = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
    timeout -k 10 200 python bench.py $Q $mode > $O/b_${hq}_${mode:2:3}.json 2>/dev/null || exit 1
    python -c "
import json,sys
d=json.load(open('$O/b_${hq}_${mode:2:3}.json'))
print('hwq=$hq mode=[$mode]', d['ms_per_step'], d['value_with_log']['ms_per_step'])"
  done
done
unset GPU_MAX_HW_QUEUES
STEP="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --no-extra-modes"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/raw_step --output-format csv -- python3 bench.py $STEP > $O/step.json 2> $O/step.err || exit 1
python3 profiles/step_trace.py $O/raw_step > $O/step_timeline.txt
head -3 $O/step_timeline.txt
export GPU_MAX_HW_QUEUES=8
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/raw_step8 --output-format csv -- python3 bench.py $STEP > $O/step8.json 2> $O/step8.err || exit 1
python3 profiles/step_trace.py $O/raw_step8 > $O/step_timeline_hwq8.txt
head -3 $O/step_timeline_hwq8.txt
rm -rf $O/raw_step $O/raw_step8
