#!/bin/bash
O=gpurun_out/r3m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.txt
if [ $rc -ge 124 ]; then exit $rc; fi
STEP="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs --no-rica --no-extra-modes"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d $O/raw_bf16_$C --output-format csv -- python3 bench.py $STEP --bf16 > /dev/null 2> $O/pmc_bf16_$C.err || exit 1
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d $O/raw_step_$C --output-format csv -- python3 bench.py $STEP > /dev/null 2> $O/pmc_step_$C.err || exit 1
done
python3 profiles/pmc_traffic.py $O/hbm_traffic.json step:$O/raw_step_FETCH_SIZE:$O/raw_step_WRITE_SIZE step_bf16:$O/raw_bf16_FETCH_SIZE:$O/raw_bf16_WRITE_SIZE
python3 -c "
import json
d=json.load(open('$O/hbm_traffic.json')); print({k:v['traffic_bytes'] for k,v in d.items() if k.startswith('step')})"
rm -rf $O/raw_*
