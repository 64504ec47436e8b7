#!/bin/bash
# round 3: chain_up at 52 VGPRs (two workgroups per CU) -- parity, then A/B against the previous build (LSHM_LIB)
set -o pipefail
O=gpurun_out/r3aa; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize_ops.py tests/test_gpu_step.py -m gpu -x -q -k "dense or chain or step" > $O/pytest.txt 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.txt
[ $rc -ne 0 ] && exit $rc
bash profiles/run_ab_lib.sh
