#!/bin/bash
# MFMA utilisation of k_dense_product (the one default kernel on MFMA: prime n+1 decks, BASELINE config 2): rocprofv3 --pmc in
# its own pass next to --kernel-trace only.  usage (GPU box, repo root): bash scripts/collect_mfma.sh <tag>
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out; mkdir -p $(dirname $OUT/${TAG}_x)
export TMPDIR=/tmp
R=$PWD
cd /tmp && rm -rf /tmp/prof_mf
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_mf -o pmc -- python $R/bench.py --config2 --steps 512 --inflight 1 > /dev/null 2>&1
python $R/scripts/pmc_sq.py $(find /tmp/prof_mf -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_pmc_mfma_config2.csv
rm -rf /tmp/prof_mf
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_mf -o kt -- python $R/bench.py --config2 --steps 512 --inflight 1 > $OUT/${TAG}_config2_bench_under_rocprof.json 2>/dev/null
python $R/scripts/kstats.py $(find /tmp/prof_mf -name "*.db" | head -1) 576 12 > $OUT/${TAG}_config2_kstats.txt
cd $R
head -4 $OUT/${TAG}_pmc_mfma_config2.csv | cut -c1-250; head -6 $OUT/${TAG}_config2_kstats.txt | cut -c1-60,100-200
