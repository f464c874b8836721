#!/bin/bash
# round 4, GPU call 9: several stages per rank with the closing edge on RCCL (one rank): parity and rate against the in-process ring
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fullsize_boxes.py -m gpu -x -q -k "closing_edge or in_flight or rccl or properties" > $O/tests9.log 2>&1; tail -4 $O/tests9.log
python bench.py --cpu-slices 0 --steps 3072 --inflight 3 > $O/plain_3boxes.json 2>> $O/b9.err
python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self > $O/ringself_3boxes.json 2>> $O/b9.err
python - <<'PY'
import json
for f in ("plain_3boxes","ringself_3boxes"):
    d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"],1), d.get("value_steps_in_flight"), d.get("ring"), d.get("rccl_ranks_seen"), d["in_flight"]["window"])
PY
tail -3 $O/b9.err
