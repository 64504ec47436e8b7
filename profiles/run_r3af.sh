#!/bin/bash
# round 3: reconstruction pass forms x2 / x3c from the input of the last 1-D layer (no-grad forward stops one layer early) -- parity, A/B, phases
set -o pipefail
O=gpurun_out/r3af; mkdir -p $O; rm -f $O/b_*.json
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py tests/test_gpu_dp.py -m gpu -x -q -k "first_1d or step or dp" > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
F="--steps 300 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2; do
timeout -k 10 300 python bench.py $F > $O/b_on_$rep.json 2> $O/b.err || exit 1
LSHM_RECON_FROM_A_OFF=1 timeout -k 10 300 python bench.py $F > $O/b_off_$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r3af/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1].rsplit('_',1)[0]].append(d['ms_per_step'])
for k,v in r.items(): print(k, v, sum(v)/len(v))
PY
LSHM_PHASE_EVENTS=1 timeout -k 10 300 python profiles/phase_times_probe.py 2>&1 | tail -12
