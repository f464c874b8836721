#!/bin/bash
# round 5, call 5: what of bench.py's sequence (headline run through the ring, then the stages) costs the 3-stage rate?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
run () { tag=$1; shift; python scripts/inflight_run.py --stages 3 --boxes 2 "$@" > $O/c5_$tag.json 2>> $O/c5.err; }
run pre_ipc --ring-self ipc --pre-pipeline 2
run pre_ipc_freshT --ring-self ipc --pre-pipeline 2 --fresh-transport
run pre_ipc_freshE --ring-self ipc --pre-pipeline 2 --fresh-engines
run pre_none --pre-pipeline 2
run pre_ipc_nowarm --ring-self ipc --pre-pipeline 2 --warm 0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c5_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["slices_per_s"], 1))
PY
