#!/bin/bash
# round 3: tconv2d_direct<12,8> in the row-parity form (K = 72), 4- or 8-row tiles -- parity, probe, step A/B
set -o pipefail
O=gpurun_out/r3ad; mkdir -p $O; rm -f $O/b_*.json
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
for v in "h4_512:LSHM_GRID_TCONV2D_12_8=512" "h4_768:LSHM_GRID_TCONV2D_12_8=768" "h4_1024:LSHM_GRID_TCONV2D_12_8=1024" "h8_512:LSHM_TCONV2D_H8=1" "h8_768:LSHM_TCONV2D_H8=1 LSHM_GRID_TCONV2D_12_8=768"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout -k 10 300 python profiles/outer2d_probe.py > $O/probe_$name.txt 2>&1; echo $name; grep -E "tconv4" $O/probe_$name.txt
done
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2; do
timeout -k 10 300 python bench.py $F > $O/b_h4_$rep.json 2> $O/b.err || exit 1
LSHM_TCONV2D_H8=1 timeout -k 10 300 python bench.py $F > $O/b_h8_$rep.json 2> $O/b.err || exit 1
LSHM_GRID_TCONV2D_12_8=768 timeout -k 10 300 python bench.py $F > $O/b_h4c768_$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3ad/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
