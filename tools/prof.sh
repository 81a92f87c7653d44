#!/bin/bash
# Profiling recipe for the GPU box: kernel-trace stats pass, then PMC passes (one rocprofv3 run each, counters only).
#   tools/prof.sh <tag> [frames]      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
FRAMES=${2:-8}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BIN=./pathtracer_amd/ptbench
if [ -z "${UBENCH_ARGS:-}" ]; then   # UBENCH_ARGS="miss_l1 miss_l2 lanes waves": only the micro-benchmark calibration at the bottom
timeout -k 10 120 $BIN $FRAMES > $OUT/plain.json || exit 1
# the same frames with PT_FLAG_COUNT_TRAVERSAL (4): node-pair visits, triangle tests and wave-iterations -> lane-steps per launch
timeout -k 10 180 $BIN $FRAMES 1920 1080 0 4 > $OUT/counted.json || exit 1
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BIN $FRAMES > $OUT/stats.log 2>&1 || exit 1
pass() { # name counters...
	local name=$1; shift
	timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BIN $FRAMES > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE GRBM_COUNT
pass tc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 tools/prof_summary.py $OUT $TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
else
echo "$UBENCH_ARGS" > $OUT/ubench_args.txt
fi
# calibration of bench.py's gather ceiling: the micro-benchmark at the kernel's per-lane-step miss rates (written by
# prof_summary.py), under the same cache counters -- its own requests per lane-step should come out at those rates
if [ -s $OUT/ubench_args.txt ] && [ -x ./pathtracer_amd/ubench_gather ]; then
	ARGS=$(cat $OUT/ubench_args.txt)
	timeout -k 10 120 ./pathtracer_amd/ubench_gather tiers $ARGS > $OUT/ubench_plain.txt || exit 1
	timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/ubench -o p -- ./pathtracer_amd/ubench_gather tiers $ARGS > $OUT/ubench.log 2>&1 || { echo "ubench pass failed"; tail -3 $OUT/ubench.log; exit 1; }
	python3 tools/prof_summary.py $OUT $TAG ubench > $OUT/${TAG}_ubench_calibration.txt 2>&1
	cat $OUT/${TAG}_ubench_calibration.txt
fi
