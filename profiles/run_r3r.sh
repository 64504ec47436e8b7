#!/bin/bash
# round 3: LDS-staged one-pass backward of the 12/8-channel 1-D layers -- parity, then A/B in the step
set -o pipefail
O=gpurun_out/r3r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -m gpu -x -q -k "one_pass or step or bf16 or overlapped" > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica > $O/b_lds$rep.json 2> $O/b.err || exit 1
LSHM_BWD_LDS_OFF=1 timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica > $O/b_off$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3r/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
