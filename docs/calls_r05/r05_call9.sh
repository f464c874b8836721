#!/bin/bash
# round 5, call 9: kernel trace of the slow case (bench.py --ring-self --inflight 3: engines of stages 1, 2 created after the ring's streams)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
for v in slow fast; do
  rm -rf /tmp/prof_if
  ex=""; [ $v = fast ] && ex="BENCH_ENGINES_FIRST=1"
  (cd /tmp && env $ex timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_if -o kt -- python $R/bench.py --cpu-slices 0 --steps 2048 --inflight 3 --ring-self --edge ipc > $O/c9_bench_$v.json 2>/dev/null)
  CSV=$(find /tmp/prof_if -name "*kernel_trace.csv" | head -1)
  python scripts/overlap_analysis.py $CSV > $O/c9_overlap_$v.txt
  python scripts/stage_phase.py $CSV > $O/c9_phase_$v.txt
  python - $CSV > $O/c9_queues_$v.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
c = collections.Counter((r.get("Queue_Id"), r.get("Stream_Id"), r.get("Thread_Id")) for r in rows)
for k, v in c.most_common(): print(k, v)
PY
done
head -8 $O/c9_overlap_slow.txt; cat $O/c9_phase_slow.txt $O/c9_queues_slow.txt; head -8 $O/c9_overlap_fast.txt; cat $O/c9_queues_fast.txt
