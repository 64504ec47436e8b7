#!/bin/bash
set -o pipefail
O=gpurun_out/r4s; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2 3; do
timeout -k 10 300 python bench.py $F > $O/b_on$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4s/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
