#!/bin/bash
# round 5, call 13: more than three stages in flight, now that the stages' hardware queues are under control (pool, 8 queues)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
run () { tag=$1; shift; env "$@" > $O/c13_$tag.json 2>> $O/c13.err; }
run 3st_q4_p3 python scripts/inflight_run.py --stages 3 --boxes 1
run 3st_q8_p3 GPU_MAX_HW_QUEUES=8 python scripts/inflight_run.py --stages 3 --boxes 1
run 4st_q8_p4 GPU_MAX_HW_QUEUES=8 HPS_STREAM_POOL=4 python scripts/inflight_run.py --stages 4 --boxes 1
run 4st_q4_p3 python scripts/inflight_run.py --stages 4 --boxes 1
run 5st_q8_p5 GPU_MAX_HW_QUEUES=8 HPS_STREAM_POOL=5 python scripts/inflight_run.py --stages 5 --boxes 1
run 6st_q8_p6 GPU_MAX_HW_QUEUES=8 HPS_STREAM_POOL=6 python scripts/inflight_run.py --stages 6 --boxes 1
run 4st_q16_p4 GPU_MAX_HW_QUEUES=16 HPS_STREAM_POOL=4 python scripts/inflight_run.py --stages 4 --boxes 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c13_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["slices_per_s"], 1))
PY
