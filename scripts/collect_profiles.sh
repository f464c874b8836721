#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel stats of the default bench run and the PMC passes (FETCH_SIZE, WRITE_SIZE, SQ
# counters: separate passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Run on the GPU box from the
# repo root:  bash scripts/collect_profiles.sh r02g      -> gpurun_out/<tag>_*
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# 1. kernel trace of the default bench command (one whole box)
rm -rf /tmp/prof_kt
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --cpu-slices 0 --inflight 1 > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/kstats_csv.py $DB > $OUT/${TAG}_kernel_stats.csv
python $GRAFT_REPO_ROOT/scripts/kstats.py $DB 1088 40 > $OUT/${TAG}_kernel_stats.txt
# 2. PMC passes on a short run (24 timed slices at the representative window; warm-up + positioning slices are counted too:
#    the per-launch averages are over all launches of a kernel)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --cpu-slices 0 --inflight 1 --steps 24 --warmup 4 --start-slice 36 > /dev/null 2>&1
  cp $(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1) /tmp/pmc_$C.csv
done
python $GRAFT_REPO_ROOT/scripts/pmc_traffic.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv > $OUT/${TAG}_pmc_fetch_write_per_kernel.csv
rm -rf /tmp/prof_pmc
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d /tmp/prof_pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --cpu-slices 0 --inflight 1 --steps 24 --warmup 4 --start-slice 36 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_sq.py $(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_sq_per_kernel.csv
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${TAG}_bench_plain.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_steps20.json 2>/dev/null
head -12 $OUT/${TAG}_kernel_stats.txt; head -6 $OUT/${TAG}_pmc_fetch_write_per_kernel.csv; head -4 $OUT/${TAG}_pmc_sq_per_kernel.csv | cut -c1-300
