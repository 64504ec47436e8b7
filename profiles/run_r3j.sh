#!/bin/bash
O=gpurun_out/r3j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize_ops.py -x -q -k "chain" > $O/pytest_chain.txt 2>&1
rc=$?; echo "chain pytest rc=$rc"; tail -3 $O/pytest_chain.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 120 python3 profiles/chain_probe.py 2>&1 | tee $O/chain_probe.txt
Q="--no-extra-modes --no-roofline --no-cpu-baseline --no-lbfgs --no-rica --no-reuse-mode --steps 40 --warmup 5"
run() { n=$1; shift
  env "$@" timeout -k 10 200 python bench.py $Q > $O/b_$n.json 2>$O/b_$n.err || { tail -3 $O/b_$n.err; exit 1; }
  python -c "
import json
d=json.load(open('$O/b_$n.json')); print('$n', d['ms_per_step'], d['value_with_log']['ms_per_step'])"
}
run chains X=1
run nochain LSHM_CHAIN_OFF=1
run chains2 X=1
