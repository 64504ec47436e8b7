#!/bin/bash
O=gpurun_out/r3p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize_ops.py -x -q -k "dense_middle" > $O/pytest_dense.txt 2>&1
rc=$?; echo "dense pytest rc=$rc"; tail -12 $O/pytest_dense.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -x -q > $O/pytest.txt 2>&1
rc=$?; echo "step pytest rc=$rc"; tail -6 $O/pytest.txt
if [ $rc -ge 124 ]; then exit $rc; fi
Q="--no-extra-modes --no-roofline --no-cpu-baseline --no-lbfgs --no-rica --no-reuse-mode --steps 40 --warmup 5"
run() { n=$1; shift
  env "$@" timeout -k 10 200 python bench.py $Q > $O/b_$n.json 2>$O/b_$n.err || { tail -3 $O/b_$n.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b_$n.json')); print('$n', d['ms_per_step'], d['value_with_log']['ms_per_step'])"
}
run dense X=1
run nodense LSHM_DENSE1D_OFF=1
run dense2 X=1
run nodense2 LSHM_DENSE1D_OFF=1
