#!/bin/bash
# round 5, call 25: node-centred multigrid with the prolongation through LDS and 64 x 32 tiles for its fused 8-sweep pass: parity, rates on the 2^K - 1 grids
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "multigrid or head_slices or poisson" 2>&1 | tail -2
for n in 1023 511 255 1024; do
  python bench.py --cpu-slices 0 --n $n > $O/c25_n$n.json 2>> $O/c25.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c25_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"), {k: round(v, 4) for k, v in d["phase_ms_per_slice"].items() if v}, d["vcycles_per_slice"])
PY
