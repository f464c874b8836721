#!/bin/bash
# round 4, GPU call 10: closing edge on RCCL with the environment a multi-rank run has (8 hardware queues, 2 RCCL channels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
GPU_MAX_HW_QUEUES=8 python bench.py --cpu-slices 0 --steps 3072 --inflight 3 > $O/plain_3boxes_q8.json 2>> $O/b10.err
GPU_MAX_HW_QUEUES=8 NCCL_MAX_P2P_NCHANNELS=2 python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self > $O/ringself_3boxes_q8c2.json 2>> $O/b10.err
GPU_MAX_HW_QUEUES=8 python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self > $O/ringself_3boxes_q8.json 2>> $O/b10.err
NCCL_MAX_P2P_NCHANNELS=2 python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self > $O/ringself_3boxes_c2.json 2>> $O/b10.err
GPU_MAX_HW_QUEUES=8 NCCL_MAX_P2P_NCHANNELS=2 python bench.py --cpu-slices 0 --steps 2048 --inflight 2 --ring-self > $O/ringself_2stages_q8c2.json 2>> $O/b10.err
python - <<'PY'
import json
for f in ("plain_3boxes_q8","ringself_3boxes_q8c2","ringself_3boxes_q8","ringself_3boxes_c2","ringself_2stages_q8c2"):
    try:
        d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"],1), d.get("value_steps_in_flight"), (d.get("ring") or {}).get("sent"), d.get("rccl_ranks_seen"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/b10.err
