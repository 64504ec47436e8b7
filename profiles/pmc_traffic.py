#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately,
as MI355X_MICROARCH.md prescribes).  gfx950 correction: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
Usage: pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json"""
import csv, glob, json, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("lshm::", "").replace("void ", "").split("(")[0]
            acc[k].append(float(r["Counter_Value"]))
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
ALGO = {  # algorithmic bytes per launch at B=256
    "tconv1d_stream_kernel<8, 4, false>": 201326592,
    "conv1d_stream_kernel<4, 8, true>": 201326592,
    "tconv2d_direct_kernel<8, 4, 4, 64>": 100663296,
    "conv2d_direct_kernel<4, 8, 4, 64>": 100663296,
    "conv2d_wgrad_direct_kernel<8, 4, 4, 64>": 100663296,
    "conv1d_wgrad_direct_kernel<8, 4, 256>": 201326592,
    "recon_kernel": 671088640,
    "multiplier_update_kernel": 671088640,
}
out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of "
                 "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` on MI355X; per-launch averages; gfx950 "
                 "correction traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B (MI355X_MICROARCH.md HBM section)",
       "kernels": {}}
for k in sorted(set(fetch) & set(write)):
    if "at::" in k:
        continue
    f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
    e = {"launches": len(fetch[k]), "FETCH_SIZE_KB": round(f), "WRITE_SIZE_KB": round(w),
         "traffic_bytes_per_launch": int((2 * f + w) * 1024)}
    if k in ALGO:
        e["algorithmic_bytes_per_launch"] = ALGO[k]
    out["kernels"][k] = e
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(out['kernels'])} kernels -> {sys.argv[3]}")
