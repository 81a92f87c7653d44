#!/bin/bash
# End-of-round evidence on the GPU box: the profiling recipe, then bench.py priced with the profile just taken, bench.py under
# rocprofv3 --kernel-trace --stats, and the other configurations.   tools/final_evidence.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-r03c}
export TMPDIR=/tmp
set -x
tools/prof.sh $TAG 32 > gpurun_out/prof_$TAG.log 2>&1 || { tail -20 gpurun_out/prof_$TAG.log; exit 1; }
cp gpurun_out/prof_$TAG/${TAG}_pmc_traffic.json profiles/ || exit 1 # for the bench runs below; profiles/ on the box is not merged back: copy it from gpurun_out/ again at home
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail gpurun_out/${TAG}_bench.err; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_benchprof -o b -- python3 bench.py --no-cpu-baseline --no-anyhit --sustain 0 > gpurun_out/${TAG}_bench_py_under_rocprof.json 2> gpurun_out/${TAG}_benchprof.err || { tail gpurun_out/${TAG}_benchprof.err; exit 1; }
timeout -k 10 200 python3 bench.py --scene bunny_ggx --no-cpu-baseline > gpurun_out/${TAG}_bench_ggx.json 2>> gpurun_out/${TAG}_bench.err || exit 1
timeout -k 10 200 python3 bench.py --width 3840 --height 2160 --no-cpu-baseline > gpurun_out/${TAG}_bench_4k.json 2>> gpurun_out/${TAG}_bench.err || exit 1
timeout -k 10 200 python3 bench.py --scene atrium --no-cpu-baseline > gpurun_out/${TAG}_bench_atrium.json 2>> gpurun_out/${TAG}_bench.err || exit 1
timeout -k 10 200 python3 bench.py --bvh ploc --no-cpu-baseline > gpurun_out/${TAG}_bench_ploc.json 2>> gpurun_out/${TAG}_bench.err || exit 1
for f in bench bench_ggx bench_4k bench_atrium bench_ploc; do python3 -c "
import json,sys; d=json.load(open('gpurun_out/${TAG}_$f.json')); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], d.get('sustained_ms_per_step'), r['bound'], r['frac'], r.get('valu_useful_frac'), (d.get('anyhit') or {}).get('value'))"; done
