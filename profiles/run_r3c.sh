#!/bin/bash
O=gpurun_out/r3c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -x -q -k "one_pass or overlapped or fused or golden or oracle or additivity" > $O/pytest.txt 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.txt
if [ $rc -ge 124 ]; then exit $rc; fi
Q="--no-extra-modes --no-roofline --no-cpu-baseline --no-lbfgs --no-rica --no-reuse-mode --steps 40 --warmup 5"
timeout -k 10 200 python bench.py $Q > $O/b.json 2>$O/b.err || { tail -5 $O/b.err; exit 1; }
LSHM_BWD_FUSED_OFF=1 timeout -k 10 200 python bench.py $Q > $O/b_nofuse.json 2>/dev/null || exit 1
python -c "
import json
for f in ('b','b_nofuse'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['ms_per_step'], d['value_with_log']['ms_per_step'])"
STEP="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --no-extra-modes"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/raw_step --output-format csv -- python3 bench.py $STEP > $O/step.json 2> $O/step.err || exit 1
python3 profiles/step_trace.py $O/raw_step > $O/step_timeline.txt
head -3 $O/step_timeline.txt
rm -rf $O/raw_step
