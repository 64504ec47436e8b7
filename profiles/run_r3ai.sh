#!/bin/bash
# round 3: bf16-storage forms of the LDS one-pass backward kernels (1-D conv0, 2-D tconv4 / conv1) -- bf16 tests, bf16 A/B
set -o pipefail
O=gpurun_out/r3ai; mkdir -p $O; rm -f $O/b_*.json
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize_ops.py tests/test_gpu_ops.py -m gpu -x -q -k "bf16 or overlapped or one_pass" > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
F="--steps 300 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --bf16"
for rep in 1 2; do
timeout -k 10 300 python bench.py $F > $O/b_on_$rep.json 2> $O/b.err || exit 1
LSHM_BWD_LDS_8_4_OFF=1 LSHM_BWD_LDS2D_OFF=1 timeout -k 10 300 python bench.py $F > $O/b_off_$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r3ai/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1].rsplit('_',1)[0]].append((d['ms_per_step'], d.get('loss_total')))
for k,v in r.items(): print(k, v)
PY
