#!/bin/bash
# round 3: ELU' inputs of the chain stages prefetched before the stage's matrix instructions -- parity, A/B against the previous build
set -o pipefail
O=gpurun_out/r3ag; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -m gpu -x -q -k "chain or step" > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python profiles/chain_probe.py > $O/chain_probe.txt 2>&1; cat $O/chain_probe.txt
bash profiles/run_ab_lib.sh
