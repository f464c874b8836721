#!/bin/bash
# round 5, call 11: two processes on one device: size of the engines' stream pool against the aggregate rate
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for P in 0 1 2 3 4; do
  HPS_STREAM_POOL=$P timeout 600 python bench.py --gpus 2 --same-device --inflight 1 --steps 20 --warmup 5 --cpu-slices 0 > $O/c11_2ranks_pool$P.json 2>> $O/c11.err
done
HPS_STREAM_POOL=0 timeout 600 python bench.py --gpus 2 --same-device --inflight 1 --steps 1024 --cpu-slices 0 > $O/c11_2ranks_pool0_wholebox.json 2>> $O/c11.err
HPS_STREAM_POOL=4 timeout 600 python bench.py --gpus 2 --same-device --inflight 1 --steps 1024 --cpu-slices 0 > $O/c11_2ranks_pool4_wholebox.json 2>> $O/c11.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c11_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d.get("value", d.get("slices_per_s")), 1), d.get("value_steps_in_flight"), d.get("ranks_seen"))
PY
