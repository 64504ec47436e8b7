#!/bin/bash
# bf16 storage: conv2d_q4 / tconv2d_bwd_fused on v_mfma_f32_4x4x4_bf16: bf16 tests, error against the oracle, A/B against the library before
set -e
O=gpurun_out/r4au; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "bf16 or precision or storage or one_pass_backward" > $O/test_bf16.txt 2>&1 || { tail -40 $O/test_bf16.txt; exit 1; }
tail -2 $O/test_bf16.txt
timeout -k 10 300 python profiles/bf16_error_probe.py > $O/bf16_err.txt 2>&1; tail -3 $O/bf16_err.txt
F="--steps 300 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py $F --bf16 > $O/bf_new$rep.json 2>$O/err.txt || { tail -5 $O/err.txt; exit 1; }
  LSHM_LIB=$PWD/build/old/liblshm_hip_pre_q4.so timeout -k 10 300 python bench.py $F --bf16 > $O/bf_old$rep.json 2>$O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
timeout -k 10 300 python bench.py $F > $O/f32_new.json 2>$O/err.txt
LSHM_LIB=$PWD/build/old/liblshm_hip_pre_q4.so timeout -k 10 300 python bench.py $F > $O/f32_old.json 2>$O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4au/*_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d['loss_total'])
PY
