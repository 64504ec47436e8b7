#!/bin/bash
# round 4: placement of the deep layers' weight gradients between the two backward streams (tune word = mask + 1)
set -o pipefail
O=gpurun_out/r4g; mkdir -p $O
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
# masks: bit0 tconv2, 1 tconv1, 2 tconv0, 3 conv5, 4 conv4, 5 conv3, 6 conv2
for m in 0 31 7 24 56 96 127 64 95 15; do
  t=$((m+1))
  timeout -k 10 300 python bench.py $F --tune $t > $O/b_m$m.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g/b_m*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
for m in 0 31; do
PROBE_TUNE=$((m+1)) timeout -k 10 120 python profiles/phase_times_probe.py > $O/phase_m$m.txt 2>&1; tail -3 $O/phase_m$m.txt
done
PROBE_SCHEDULE_OFF=no_deep2d_bwd timeout -k 10 120 python profiles/phase_times_probe.py > $O/phase_nobwd.txt 2>&1; tail -3 $O/phase_nobwd.txt
