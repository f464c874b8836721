#!/bin/bash
# round 4, GPU call 20: the L-stages-in-flight window of the shipped (r04g) build -- L sweep, kernel overlap, FETCH / WRITE counters over the 3-stage window
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for L in 1 2 3 4; do python scripts/inflight_run.py --stages $L --boxes 1 2>/dev/null | tail -1; done > $O/r04g_inflight_sweep.jsonl
cat $O/r04g_inflight_sweep.jsonl | cut -c1-200
cd /tmp; rm -rf /tmp/prof_if
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_if -o kt -- python $R/scripts/inflight_run.py --stages 3 --boxes 1 > $O/r04g_inflight3_under_trace.json 2>/dev/null
python $R/scripts/overlap_analysis.py $(find /tmp/prof_if -name "*kernel_trace.csv" | head -1) > $O/r04g_inflight_overlap_3_stages.txt
head -12 $O/r04g_inflight_overlap_3_stages.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  timeout 1200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -o pmc -- python $R/scripts/inflight_run.py --stages 3 --boxes 1 --warm 0 > $O/r04g_inflight3_under_pmc_$C.json 2>/dev/null
  cp $(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1) /tmp/ifpmc_$C.csv
done
python $R/scripts/pmc_traffic.py /tmp/ifpmc_FETCH_SIZE.csv /tmp/ifpmc_WRITE_SIZE.csv > $O/r04g_inflight_pmc.csv
head -6 $O/r04g_inflight_pmc.csv | cut -c1-160
cd $R; python - <<'PY'
import csv
tot = 0.0; n = 0
for r in csv.DictReader(open("gpurun_out/r04/r04g_inflight_pmc.csv")):
    try: b = (2.0*float(r["FETCH_SIZE_raw_per_launch"]) + float(r["WRITE_SIZE_raw_per_launch"]))*1024.0
    except ValueError: continue
    if b != b: continue
    tot += b*int(r["launches"])
    if r["kernel"].startswith("void hps::k_deposit_tiled<2, 16, 51"): n += int(r["launches"])
print("three-stage window: %.3f GB per slice over %d slices" % (tot/max(n,1)/1e9, n))
PY
