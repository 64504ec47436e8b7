#!/bin/bash
# round 3: pipelined conv2d_direct (4 x 32 tiles, 512 persistent workgroups), tconv2d_direct with 512 workgroups -- parity + step
set -o pipefail
O=gpurun_out/r3y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python profiles/outer2d_probe.py > $O/outer2d_probe.txt 2>&1; cat $O/outer2d_probe.txt
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2; do
timeout -k 10 300 python bench.py $F > $O/b_new$rep.json 2> $O/b.err || exit 1
LSHM_GRID_TCONV2D_12_8=768 timeout -k 10 300 python bench.py $F > $O/b_t768_$rep.json 2> $O/b.err || exit 1
done
timeout -k 10 300 python bench.py $F --bf16 > $O/b_bf16.json 2> $O/b.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3y/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
