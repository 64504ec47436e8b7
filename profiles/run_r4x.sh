#!/bin/bash
# round 4 experiment: the closure forward's chain held L steps behind the no-grad forward's (gate = one cross-stream event), one or two patches per deep workgroup
set -e
O=gpurun_out/r4x; mkdir -p $O
F="--steps 300 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2; do
for tune in 0 2097152 196608 1179648 1245184 3342336 1310720 3407872 1376256 3473408 1441792 1572864; do
  timeout -k 10 300 python bench.py $F --tune $tune > $O/b_${tune}_$rep.json 2>$O/err.txt || { tail -5 $O/err.txt; exit 1; }
done; done
python - <<'PY'
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r4x/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=int(f.split('b_')[1].split('_')[0]); r[t].append(d['ms_per_step'])
for t,v in sorted(r.items()): print(f"tune={t:8d} lag={(t>>16)&15} gate={(t>>20)&1} g1={(t>>21)&1}  ", v)
PY
