#!/bin/bash
# the whole GPU suite, then the default bench line
set -o pipefail
O=gpurun_out/full; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?
echo "bench rc=$rc"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/full/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'])
for k in ('sequential_forwards_mode','bf16_mode','k64_mode','admm10_loop','reuse_forward_mode','lbfgs_iteration'):
    print(k, d.get(k))
PY
