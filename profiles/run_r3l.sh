#!/bin/bash
O=gpurun_out/r3l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -x -q -k "bf16" > $O/pytest_bf16.txt 2>&1
rc=$?; echo "bf16 pytest rc=$rc"; tail -25 $O/pytest_bf16.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 100 python profiles/bf16_error_probe.py > $O/bf16_error.txt 2>&1; tail -12 $O/bf16_error.txt
Q="--no-extra-modes --no-roofline --no-cpu-baseline --no-lbfgs --no-rica --no-reuse-mode --steps 40 --warmup 5"
timeout -k 10 200 python bench.py $Q --bf16 > $O/b_bf16.json 2>$O/b_bf16.err || { tail -5 $O/b_bf16.err; exit 1; }
python -c "
import json
d=json.load(open('$O/b_bf16.json')); print('bf16', d['ms_per_step'], d['value_with_log']['ms_per_step'], d['loss_total'])"
