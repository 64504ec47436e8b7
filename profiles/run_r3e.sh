#!/bin/bash
O=gpurun_out/r3e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
Q="--no-extra-modes --no-roofline --no-cpu-baseline --no-lbfgs --no-rica --no-reuse-mode --steps 40 --warmup 5"
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 200 python bench.py $Q > $O/b_$n.json 2>$O/b_$n.err || { tail -3 $O/b_$n.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b_$n.json')); print('$n', d['ms_per_step'], d['value_with_log']['ms_per_step'])"
}
run nofuse LSHM_BWD_FUSED_OFF=1
run fused_default X=1
run fused_no2d LSHM_BWD_FUSED2D_OFF=1
run fused_nosplit LSHM_TUNE_FILE=$GRAFT_REPO_ROOT/profiles/tuned_nosplit_fwd.txt
run seq_nofuse LSHM_BWD_FUSED_OFF=1 LSHM_FORWARD_STAGGER=100
STEP="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --no-extra-modes"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/raw_step --output-format csv -- python3 bench.py $STEP > $O/step.json 2> $O/step.err || exit 1
python3 profiles/step_trace.py $O/raw_step > $O/step_timeline.txt
head -2 $O/step_timeline.txt
rm -rf $O/raw_step
