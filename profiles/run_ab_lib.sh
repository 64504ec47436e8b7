#!/bin/bash
# A/B of two builds of the library on one box: lshm_amd/lib/liblshm_hip.so (new) against liblshm_base.so (LSHM_LIB)
set -o pipefail
O=gpurun_out/ab; mkdir -p $O; rm -f $O/b_*.json
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2 3; do
timeout -k 10 300 python bench.py $F > $O/b_new$rep.json 2> $O/b.err || exit 1
LSHM_LIB=$PWD/lshm_amd/lib/liblshm_base.so timeout -k 10 300 python bench.py $F > $O/b_base$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
