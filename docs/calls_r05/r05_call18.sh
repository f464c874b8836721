#!/bin/bash
# round 5, call 18: the power-of-two DST kernel (2^K - 1 cells per side): parity, rate at n = 1023 / 511
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "poisson or head_slices" 2>&1 | tail -4 > $O/c18_tests.txt
cat $O/c18_tests.txt
python bench.py --cpu-slices 0 --n 1023 --inflight 1 > $O/c18_n1023.json 2>> $O/c18.err
python bench.py --cpu-slices 0 --n 511 --inflight 1 > $O/c18_n511.json 2>> $O/c18.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c18_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), {k: (round(v, 4) if v else v) for k, v in d["phase_ms_per_slice"].items()}, d["vcycles_per_slice"])
PY
