#!/bin/bash
O=gpurun_out/r3v; mkdir -p $O
timeout -k 10 300 python profiles/outer2d_probe.py > $O/outer2d_probe.txt 2>&1 || { tail -5 $O/outer2d_probe.txt; exit 1; }
cat $O/outer2d_probe.txt
