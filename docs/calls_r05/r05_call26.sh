#!/bin/bash
# round 5, call 26: the current build against the r05a build on ONE box (the multigrid source changed for the node-centred path only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for i in 1 2; do
  HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_r05a.so python bench.py --cpu-slices 0 > $O/c26_r05a_$i.json 2>> $O/c26.err
  python bench.py --cpu-slices 0 > $O/c26_now_$i.json 2>> $O/c26.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c26_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"), round(d["phase_ms_per_slice"]["mg_solve1"], 4))
PY
