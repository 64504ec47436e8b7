#!/bin/bash
set -o pipefail
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_ops.py -m gpu -x -q -k "deep_section or whole_1d" > $O/pytest_ops.txt 2>&1; rc=$?
echo "pytest ops rc=$rc"; tail -3 $O/pytest_ops.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python profiles/deep2d_probe.py > $O/probe.txt 2>&1; cat $O/probe.txt
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2; do
timeout -k 10 300 python bench.py $F > $O/b_on$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4m/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
