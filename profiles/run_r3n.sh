#!/bin/bash
O=gpurun_out/r3n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in 256 512 768 1024; do echo "tconv2d<12,8> grid cap $g"; LSHM_GRID_TCONV2D_12_8=$g timeout 60 python profiles/layer_bench.py 256 1 4 2>/dev/null; done | tee $O/tconv4.txt
for g in 256 512 768 1024; do echo "conv2d<8,12> grid cap $g"; LSHM_GRID_CONV2D_8_12=$g timeout 60 python profiles/layer_bench.py 256 0 1 2>/dev/null; done | tee $O/conv1.txt
Q="--no-extra-modes --no-roofline --no-cpu-baseline --no-lbfgs --no-rica --no-reuse-mode --steps 40 --warmup 5"
timeout -k 10 200 python bench.py $Q > $O/b.json 2>/dev/null && python -c "
import json
d=json.load(open('$O/b.json')); print('step', d['ms_per_step'], d['value_with_log']['ms_per_step'])"
