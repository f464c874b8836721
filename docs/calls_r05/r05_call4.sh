#!/bin/bash
# round 5, call 4: does last round's 3-stage loss on the ring (bench.py --ring-self --inflight 3 --steps 3072) reproduce?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 NCCL_MAX_P2P_NCHANNELS=2
for edge in rccl ipc; do
  python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self --edge $edge > $O/c4_bench_ringself_3st_$edge.json 2>> $O/c4.err
  python scripts/inflight_run.py --stages 3 --boxes 3 --ring-self $edge > $O/c4_lanes_3st_3boxes_$edge.json 2>> $O/c4.err
done
python bench.py --cpu-slices 0 --steps 3072 --inflight 3 > $O/c4_bench_inprocess_3st.json 2>> $O/c4.err
python scripts/inflight_run.py --stages 3 --boxes 3 > $O/c4_lanes_3st_3boxes_none.json 2>> $O/c4.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c4_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d.get("value", d.get("slices_per_s")), 1), d.get("value_steps_in_flight"))
PY
