#!/bin/bash
# round 4: bf16-storage forms of the two tile kernels: parity, bf16 step A/B, the whole step + precision test files
set -e
O=gpurun_out/r4ab; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_ops.py -q -x -k "bf16_storage or first_1d_layers or reconstruction_pass" > $O/test.txt 2>&1 || { tail -40 $O/test.txt; exit 1; }
tail -1 $O/test.txt
F="--steps 300 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --bf16"
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py $F > $O/b_new$rep.json 2>$O/err.txt || { tail -5 $O/err.txt; exit 1; }
  timeout -k 10 300 python bench.py $F --schedule-off no_recon_bwd5,no_conv0_bwd_tile > $O/b_off$rep.json 2>$O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4ab/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['loss_total'])
PY
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "bf16 or precision or storage" > $O/test_bf16.txt 2>&1 || { tail -40 $O/test_bf16.txt; exit 1; }
tail -2 $O/test_bf16.txt
