#!/bin/bash
# Runs exp/<name>/ptbench for every name given (GPU box), twice each, one JSON line per run into gpurun_out/exp.log
#   tools/exp_run.sh <frames> <name>...
FRAMES=$1; shift
mkdir -p gpurun_out
: > gpurun_out/exp.log
for n in "$@"; do
	for rep in 1 2; do
		echo -n "$n " >> gpurun_out/exp.log
		timeout -k 5 45 ./exp/$n/ptbench $FRAMES >> gpurun_out/exp.log 2>&1 || { echo "FAILED $n" >> gpurun_out/exp.log; cat gpurun_out/exp.log; exit 1; }
	done
done
cat gpurun_out/exp.log
