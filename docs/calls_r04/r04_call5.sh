#!/bin/bash
# round 4, GPU call 5: the L-stages-in-flight window -- L sweep, kernel overlap, FETCH / WRITE counters; r04c profiles of the shipped build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "schedules" > $O/sched_tests.log 2>&1; tail -2 $O/sched_tests.log
for L in 1 2 3 4 5; do python scripts/inflight_run.py --stages $L --boxes 1 2>/dev/null | tail -1; done > $O/inflight_sweep.jsonl
cat $O/inflight_sweep.jsonl
cd /tmp; rm -rf /tmp/prof_if
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_if -o kt -- python $R/scripts/inflight_run.py --stages 3 --boxes 1 > $O/inflight3_under_trace.json 2>/dev/null
python $R/scripts/overlap_analysis.py $(find /tmp/prof_if -name "*kernel_trace.csv" | head -1) > $O/inflight3_overlap.txt
head -12 $O/inflight3_overlap.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  timeout 1200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -o pmc -- python $R/scripts/inflight_run.py --stages 3 --boxes 1 --warm 0 > $O/inflight3_under_pmc_$C.json 2>/dev/null
  cp $(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1) /tmp/ifpmc_$C.csv
done
python $R/scripts/pmc_traffic.py /tmp/ifpmc_FETCH_SIZE.csv /tmp/ifpmc_WRITE_SIZE.csv > $O/inflight3_pmc_fetch_write_per_kernel.csv
head -8 $O/inflight3_pmc_fetch_write_per_kernel.csv | cut -c1-160
cd $R
bash scripts/collect_profiles.sh r04/r04c > $O/collect_r04c.log 2>&1; tail -12 $O/collect_r04c.log | cut -c1-200
