#!/bin/bash
# round 3: LDS-staged one-pass backward for conv0 of the 1-D autoencoders -- parity, probe, A/B
set -o pipefail
O=gpurun_out/r3ab; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -m gpu -x -q -k "one_pass or step" > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python profiles/fused_bwd_probe.py > $O/fused_bwd_probe.txt 2>&1; cat $O/fused_bwd_probe.txt
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2 3; do
timeout -k 10 300 python bench.py $F > $O/b_lds$rep.json 2> $O/b.err || exit 1
LSHM_BWD_LDS_8_4_OFF=1 timeout -k 10 300 python bench.py $F > $O/b_reg$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3ab/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
