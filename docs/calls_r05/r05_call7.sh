#!/bin/bash
# round 5, call 7: bisect bench.py --ring-self --inflight 3, second round: creation order of the engines, headline run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
run () { tag=$1; shift; env "$@" python bench.py --cpu-slices 0 --steps 2048 --inflight 3 --ring-self --edge ipc > $O/c7_$tag.json 2>> $O/c7.err; }
run engines_first BENCH_ENGINES_FIRST=1
run skip_headline BENCH_SKIP_HEADLINE=1
run both BENCH_ENGINES_FIRST=1 BENCH_SKIP_HEADLINE=1
python scripts/inflight_run.py --stages 3 --boxes 2 --ring-self ipc --warm 64 > $O/c7_lanes_warm64.json 2>> $O/c7.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c7_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d.get("value", d.get("slices_per_s")), 1), d.get("value_steps_in_flight"))
PY
