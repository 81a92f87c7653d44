#!/bin/bash
# A/B counters for exp/<name>/ variants (GPU box): a counted replay (PT_FLAG_COUNT_TRAVERSAL) and one SQ PMC pass per variant,
# summed over the k_trace launches of a 32-frame batch.   tools/exp_pmc.sh <name>...   -> gpurun_out/exp_pmc.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/exp_pmc.txt
for n in "$@"; do
	echo "== $n" >> gpurun_out/exp_pmc.txt
	timeout -k 5 60 ./exp/$n/ptbench 32 1920 1080 0 4 >> gpurun_out/exp_pmc.txt 2>&1 || exit 1
	rm -rf gpurun_out/exp_pmc_$n
	timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/exp_pmc_$n -o p -- ./exp/$n/ptbench 32 > gpurun_out/exp_pmc_$n.log 2>&1 || { tail -3 gpurun_out/exp_pmc_$n.log; exit 1; }
	timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/exp_pmc2_$n -o p -- ./exp/$n/ptbench 32 > gpurun_out/exp_pmc2_$n.log 2>&1 || { tail -3 gpurun_out/exp_pmc2_$n.log; exit 1; }
	python3 - "$n" >> gpurun_out/exp_pmc.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
n = sys.argv[1]
a = defaultdict(float); c = defaultdict(int)
for d in ("exp_pmc_", "exp_pmc2_"):
    for f in glob.glob(f"gpurun_out/{d}{n}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_trace" in row["Kernel_Name"]:
                a[row["Counter_Name"]] += float(row["Counter_Value"]); c[row["Counter_Name"]] += 1
for k in sorted(a):
    print(f"   {k:26s} {a[k]:18.0f} over {c[k]} k_trace dispatches")
if a.get("SQ_ACTIVE_INST_VALU"):
    print(f"   VALU lane utilisation {a['SQ_THREAD_CYCLES_VALU'] / (64 * a['SQ_ACTIVE_INST_VALU']):.3f}")
PY
	rm -rf gpurun_out/exp_pmc_$n gpurun_out/exp_pmc2_$n
done
cat gpurun_out/exp_pmc.txt
