#!/bin/bash
# several stages per rank with the closing edge on the ring: which part of the ring path costs the three-stage rate?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 NCCL_MAX_P2P_NCHANNELS=2
for f in 2 0; do
  HPS_EVENT_FENCE=$f python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self > $O/ringdiag_fence$f.json 2>> $O/ringdiag.err
done
HPS_EVENT_FENCE=2 HPS_RING_SELF_COPY=1 python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self > $O/ringdiag_fence2_copy.json 2>> $O/ringdiag.err
HPS_DRIVE_TRACE=$PWD/$O/ringdiag_trace.txt python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self > $O/ringdiag_trace.json 2>> $O/ringdiag.err
python - <<'PY'
import json
for f in ("ringdiag_fence2","ringdiag_fence0","ringdiag_fence2_copy","ringdiag_trace"):
    try:
        d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"],1), d.get("value_steps_in_flight"))
    except Exception as e: print(f,"ERR",e)
PY
head -c 3000 $O/ringdiag_trace.txt
