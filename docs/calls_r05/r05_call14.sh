#!/bin/bash
# round 5, call 14: the driver's multi-GPU command lines with every rank on device 0 (N = 2, 4, 8: the N-rank code path of bench.py on the one GPU there is)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for N in 2 4 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 20 --warmup 5 --same-device > $O/c14_N${N}_same_device.json 2> $O/c14_N${N}.err
  echo "N=$N rc=$?"
done
timeout 900 python bench.py --gpus 4 --same-device --steps 1024 --cpu-slices 0 --one-stage-per-rank > $O/c14_N4_wholebox.json 2> $O/c14_N4_wholebox.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c14_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"), "ranks_seen", d.get("ranks_seen"), (d.get("in_flight") or {}).get("error"), d["timed_slices"])
PY
tail -3 $O/c14_N8.err
