#!/bin/bash
O=gpurun_out/dbg; mkdir -p $O; rm -f $O/b_*.json
F="--no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2; do
timeout -k 10 300 python bench.py $F --steps 20 --warmup 5 > $O/b_s20w5_$rep.json 2> $O/b.err || exit 1
timeout -k 10 300 python bench.py $F --steps 20 --warmup 50 > $O/b_s20w50_$rep.json 2> $O/b.err || exit 1
timeout -k 10 300 python bench.py $F --steps 200 --warmup 5 > $O/b_s200w5_$rep.json 2> $O/b.err || exit 1
timeout -k 10 300 python bench.py $F --steps 200 --warmup 50 > $O/b_s200w50_$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/dbg/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
