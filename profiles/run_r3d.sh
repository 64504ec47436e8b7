#!/bin/bash
O=gpurun_out/r3d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -x -q -k "one_pass or overlapped or fused or golden or additivity" > $O/pytest.txt 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 120 python profiles/fused_bwd_probe.py 2>&1 | tee $O/probe.txt
Q="--no-extra-modes --no-roofline --no-cpu-baseline --no-lbfgs --no-rica --no-reuse-mode --steps 40 --warmup 5"
for sg in 0 6 12 18 24; do
  LSHM_FORWARD_STAGGER=$sg LSHM_BWD_FUSED_OFF=1 timeout -k 10 200 python bench.py $Q > $O/b_s$sg.json 2>/dev/null || exit 1
  python -c "
import json
d=json.load(open('$O/b_s$sg.json')); print('stagger $sg nofuse', d['ms_per_step'], d['value_with_log']['ms_per_step'])"
done
timeout -k 10 200 python bench.py $Q > $O/b_fused.json 2>/dev/null || exit 1
python -c "
import json
d=json.load(open('$O/b_fused.json')); print('stagger 12 fused', d['ms_per_step'], d['value_with_log']['ms_per_step'])"
