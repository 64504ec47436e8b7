#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md prescribes).  gfx950 correction: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.

Every profiled command gets a label and its kernels are keyed "<kernel> @<label>", so a kernel that runs at two
shapes (the K-harmonic kernel: B=256 inside the step, N=2^20 in the roofline demonstration) is never averaged
across them.  The label `step` (bench.py --no-roofline ...) also yields the traffic of one whole ADMM iteration:
all bytes of the run divided by its number of iterations (= adam_kernel launches).

Usage: pmc_traffic.py OUT.json LABEL:FETCH_DIR:WRITE_DIR [LABEL:FETCH_DIR:WRITE_DIR ...]"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from names import short  # noqa: E402


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            acc[k].append(float(r["Counter_Value"]))
    return acc


out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes per labelled command on "
                 "MI355X; per-launch averages; gfx950 correction traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B "
                 "(MI355X_MICROARCH.md, HBM section)",
       "kernels": {}}
for spec in sys.argv[2:]:
    label, fdir, wdir = spec.split(":")
    fetch, write = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    tot = 0.0
    for k in sorted(set(fetch) & set(write)):
        f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
        if "at::" not in k and "rocclr" not in k:
            out["kernels"][f"{k} @{label}"] = {"launches": len(fetch[k]), "FETCH_SIZE_KB": round(f), "WRITE_SIZE_KB": round(w),
                                               "traffic_bytes_per_launch": int((2 * f + w) * 1024)}
        tot += (2 * sum(fetch[k]) * len(write[k]) / len(fetch[k]) + sum(write[k])) * 1024 if k in write else 0.0
    if label.startswith("step"):  # "step", "step_bf16", ...: one whole ADMM iteration of that configuration
        nsteps = len(fetch.get("adam_kernel", [])) or 1
        out[label] = {"iterations": nsteps, "traffic_bytes": int(tot / nsteps),
                       "note": "every launch of the run (warm-up included) / adam_kernel launches"}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(f"{len(out['kernels'])} kernel entries -> {sys.argv[1]}")
