#!/bin/bash
# round 4: reconstruction pass with the backward of the 1-D autoencoders' last layer inside: parity, step A/B, step tests
set -e
O=gpurun_out/r4aa; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_ops.py -q -x -k "reconstruction_pass" > $O/test.txt 2>&1 || { tail -40 $O/test.txt; exit 1; }
tail -1 $O/test.txt
F="--steps 300 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py $F > $O/b_new$rep.json 2>$O/err.txt || { tail -5 $O/err.txt; exit 1; }
  timeout -k 10 300 python bench.py $F --schedule-off no_recon_bwd5 > $O/b_off$rep.json 2>$O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4aa/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -q -x > $O/test_step.txt 2>&1 || { tail -40 $O/test_step.txt; exit 1; }
tail -2 $O/test_step.txt
TRACE_ENDS_ONLY=1 timeout -k 10 300 python profiles/step_trace_unprofiled.py > $O/completions.txt 2>$O/err2.txt || { tail -5 $O/err2.txt; exit 1; }
