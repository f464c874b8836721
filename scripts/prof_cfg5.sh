#!/bin/bash
# config 5 (laser + ionisable dopant) under rocprofv3 --kernel-trace: usage bash scripts/prof_cfg5.sh <tag> [bench args]
set -u
TAG=${1:-c5}; shift || true
OUT=$PWD/gpurun_out; mkdir -p $(dirname $OUT/${TAG}_x)
export TMPDIR=/tmp
R=$PWD
cd /tmp && rm -rf /tmp/prof_c5
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_c5 -o kt -- python $R/bench.py --config5 --cpu-slices 0 --inflight 1 --steps 512 --warmup 32 "$@" > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
DB=$(find /tmp/prof_c5 -name "*.db" | head -1)
python $R/scripts/kstats.py $DB 544 60 > $OUT/${TAG}_kstats.txt
cd $R
cut -c1-80,100-200 $OUT/${TAG}_kstats.txt | head -${QP_LINES:-45}
