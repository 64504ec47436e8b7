#!/bin/bash
# round 3: weight-gradient release group size again, now that a release costs the data-gradient stream nothing
set -o pipefail
O=gpurun_out/r3x; mkdir -p $O
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2; do
for g in 1 2 3 4; do
LSHM_WGRAD_GROUP=$g timeout -k 10 300 python bench.py $F > $O/b_g${g}_$rep.json 2> $O/b.err || exit 1
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3x/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
