#!/bin/bash
# round 5, call 19: pow2 DST with both y passes in one launch: parity, kernel stats at n = 1023
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "poisson or head_slices" 2>&1 | tail -3 > $O/c19_tests.txt
cat $O/c19_tests.txt
QP_LINES=24 bash scripts/quick_prof.sh r05/c19_n1023 --n 1023 > /dev/null 2>&1
python bench.py --cpu-slices 0 --n 1023 --inflight 3 > $O/c19_n1023.json 2>> $O/c19.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c19_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"), {k: (round(v, 4) if v else v) for k, v in d["phase_ms_per_slice"].items()}, d["vcycles_per_slice"])
PY
cut -c1-100,100-200 $O/c19_n1023_kstats.txt | head -24
