#!/bin/bash
# round 5, call 10: with the engines' stream pool: bench.py --ring-self --inflight 3 (the slow case of calls 4-9), both edges; 2 processes on one device x 1..3 stages
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for edge in ipc rccl; do
  python bench.py --cpu-slices 0 --steps 2048 --inflight 3 --ring-self --edge $edge > $O/c10_ringself_3st_$edge.json 2>> $O/c10.err
done
HPS_STREAM_POOL=0 python bench.py --cpu-slices 0 --steps 2048 --inflight 3 --ring-self --edge ipc > $O/c10_ringself_3st_ipc_nopool.json 2>> $O/c10.err
python bench.py --cpu-slices 0 --steps 2048 --inflight 3 > $O/c10_inprocess_3st.json 2>> $O/c10.err
for L in 1 2 3; do
  timeout 600 python bench.py --gpus 2 --same-device --inflight-ring --inflight $L --steps 20 --warmup 5 --cpu-slices 0 > $O/c10_2ranks_same_device_${L}st.json 2>> $O/c10.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c10_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d.get("value", d.get("slices_per_s")), 1), d.get("value_steps_in_flight"), d.get("ranks_seen"))
PY
tail -3 $O/c10.err
