#!/bin/bash
# One whole box of the default bench under rocprofv3 --kernel-trace; per-kernel averages per slice -> gpurun_out/<tag>_kstats.txt
# usage (GPU box, repo root): bash scripts/quick_prof.sh <tag> [extra bench.py arguments]
set -u
TAG=${1:-q}; shift || true
OUT=$PWD/gpurun_out; mkdir -p $(dirname $OUT/${TAG}_x)
export TMPDIR=/tmp
R=$PWD
cd /tmp && rm -rf /tmp/prof_q
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_q -o kt -- python $R/bench.py --cpu-slices 0 --inflight 1 "$@" > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
DB=$(find /tmp/prof_q -name "*.db" | head -1)
python $R/scripts/kstats.py $DB 1088 40 > $OUT/${TAG}_kstats.txt
python $R/scripts/kstats_csv.py $DB > $OUT/${TAG}_kernel_stats.csv
cd $R
cut -c1-70,100-200 $OUT/${TAG}_kstats.txt | head -${QP_LINES:-32}
