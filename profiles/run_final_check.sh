#!/bin/bash
# final check of the tree: build entry, smoke, the whole GPU suite, the driver's bench command
set -o pipefail
O=gpurun_out/final; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; rc=$?; tail -2 $O/smoke.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?
echo "bench rc=$rc"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(d['roofline']); print(d['cpu_baseline'])
PY
