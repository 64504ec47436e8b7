#!/bin/bash
# tile-table entries for B = 128 and B = 512 (the committed table was measured at B = 256 only):
# measuring mode times the 22 tile configurations of every GEMM shape of the run once (bench.py --save-tuning)
set -o pipefail
O=gpurun_out/tune; mkdir -p $O
F="--no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for B in 128 512; do
  timeout -k 10 500 python bench.py --batch $B --steps 5 --warmup 2 $F --save-tuning $O/t$B.txt > $O/tune_$B.json 2> $O/tune_$B.err || { tail -5 $O/tune_$B.err; exit 1; }
  wc -l $O/t$B.txt
done
# before / after: the static heuristic for the unknown shapes (the committed table) against the table + the new entries
cat lshm_amd/tuned_gfx950.txt $O/t128.txt $O/t512.txt | sort -n -u > $O/merged.txt
wc -l $O/merged.txt
for B in 128 512; do
  timeout -k 10 300 python bench.py --batch $B --steps 100 --warmup 20 $F > $O/b_${B}_old.json 2> $O/b.err || exit 1
  LSHM_TUNE_FILE=$PWD/$O/merged.txt timeout -k 10 300 python bench.py --batch $B --steps 100 --warmup 20 $F > $O/b_${B}_new.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/tune/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['value'])
PY
