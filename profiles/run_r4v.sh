#!/bin/bash
# round 4: conv0 of netT / netF from one 64 x 64 tile (both forms): parity, isolated timing, step A/B on one box
set -e
O=gpurun_out/r4v; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_ops.py -q -x -k "first_1d_layers" > $O/test.txt 2>&1 || { tail -30 $O/test.txt; exit 1; }
tail -3 $O/test.txt
timeout -k 10 300 python profiles/resid_conv0_probe.py > $O/probe.txt 2>&1; cat $O/probe.txt
for rep in 1 2; do
for off in "" "no_resid_conv0_keep" "no_resid_conv0,no_resid_conv0_keep"; do
  timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica ${off:+--schedule-off $off} > $O/bench_${off:-default}_$rep.json 2>$O/err.txt || { tail -5 $O/err.txt; exit 1; }
  python - "$O/bench_${off:-default}_$rep.json" "${off:-default}" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["ms_per_step"], d["value"])
PY
done; done
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -q -x > $O/test_step.txt 2>&1 || { tail -30 $O/test_step.txt; exit 1; }
tail -3 $O/test_step.txt
