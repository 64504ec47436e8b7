#!/bin/bash
# usage: bash gpurun_out/run_r3a.sh  (on the GPU box, from the repo root)
O=gpurun_out/r3k
mkdir -p $O
CAP="tests/test_gpu_dp.py::test_captured_closure_with_communicator_sends_no_early_bucket"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect "$CAP" > $O/pytest.txt 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.txt
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python -m pytest "$CAP" -q > $O/capture.txt 2>&1
rc2=$?; echo "capture rc=$rc2"; tail -3 $O/capture.txt
if [ $rc2 -ge 124 ]; then exit $rc2; fi
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
rc3=$?; echo "bench rc=$rc3"; tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3k/bench.json'))
for k in ('value','ms_per_step','value_with_log','reuse_forward_mode','sequential_forwards_mode','bf16_mode','k64_mode','admm10_loop','lbfgs_iteration'):
    print(k, d.get(k))
PY
