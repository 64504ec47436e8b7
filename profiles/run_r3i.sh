#!/bin/bash
O=gpurun_out/r3i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 120 python3 profiles/chain_probe.py 2>&1 | tee $O/chain_probe.txt
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace -d $O/raw_$n --output-format csv -- python3 profiles/chain_probe.py > /dev/null 2> $O/pmc_$n.err || { tail -3 $O/pmc_$n.err; continue; }
  python3 - "$O/raw_$n" >> $O/chain_counters.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "chain" in k or "igemm" in k:
            k = k.split("(")[0].replace("void lshm::", "")[:70]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
  rm -rf $O/raw_$n
done
cat $O/chain_counters.txt | grep chain
