#!/bin/bash
set -o pipefail
O=gpurun_out/r4r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_ops.py -m gpu -x -q -k "deep_section" > $O/pytest_ops.txt 2>&1; rc=$?
echo "pytest ops rc=$rc"; tail -3 $O/pytest_ops.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "bf16 or full_size" > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2; do
timeout -k 10 300 python bench.py $F --bf16 > $O/b_bf16_on$rep.json 2> $O/b.err || exit 1
timeout -k 10 300 python bench.py $F --bf16 --schedule-off no_deep2d,no_deep2d_bwd > $O/b_bf16_off$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4r/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
timeout -k 10 300 python profiles/bf16_error_probe.py > $O/bf16_err.txt 2>&1; tail -12 $O/bf16_err.txt
