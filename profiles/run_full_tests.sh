#!/bin/bash
# the whole GPU suite + smoke, one process each
set -e
O=gpurun_out/full; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/test_gpu.txt 2>&1 || { tail -60 $O/test_gpu.txt; exit 1; }
tail -3 $O/test_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -1 $O/smoke.txt
