#!/bin/bash
set -o pipefail
O=gpurun_out/r4u; mkdir -p $O
F="--steps 300 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2 3; do
timeout -k 10 300 python bench.py $F > $O/b_side$rep.json 2> $O/b.err || exit 1
timeout -k 10 300 python bench.py $F --tune 257 > $O/b_main$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4u/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
