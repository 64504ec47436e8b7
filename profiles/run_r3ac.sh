#!/bin/bash
# round 3: one-pass backward of the 12/8-channel 2-D layers (conv2d_fused.hip) -- parity, A/B, timeline
set -o pipefail
O=gpurun_out/r3ac; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -m gpu -x -q -k "one_pass or step" > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
F="--steps 200 --warmup 30 --no-extra-modes --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica"
for rep in 1 2 3; do
timeout -k 10 300 python bench.py $F > $O/b_on$rep.json 2> $O/b.err || exit 1
LSHM_BWD_LDS2D_OFF=1 timeout -k 10 300 python bench.py $F > $O/b_off$rep.json 2> $O/b.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3ac/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEP="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --no-extra-modes"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/raw_step --output-format csv -- python3 bench.py $STEP > $O/step.json 2> $O/step.err || exit 1
python3 profiles/step_trace.py $O/raw_step > $O/step_timeline.txt
head -1 $O/step_timeline.txt; grep "conv2d_bwd_lds\|tconv2d_bwd_fused" $O/step_timeline.txt | cut -c1-120
rm -rf $O/raw_step
