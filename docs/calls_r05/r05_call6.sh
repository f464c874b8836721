#!/bin/bash
# round 5, call 6: bisect bench.py --ring-self --inflight 3: which part of its sequence costs the 3-stage rate on the ring?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
run () { tag=$1; shift; env "$@" python bench.py --cpu-slices 0 --steps 2048 --inflight 3 --ring-self --edge ipc > $O/c6_$tag.json 2>> $O/c6.err; }
run base A=1
run freshT BENCH_LANES_FRESH_TRANSPORT=1
run freshE BENCH_LANES_FRESH_ENGINES=1
run noprof BENCH_NO_PROFILING=1
run trace HPS_DRIVE_TRACE=$PWD/$O/c6_hosttrace.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c6_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"))
PY
