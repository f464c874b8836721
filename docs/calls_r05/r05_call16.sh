#!/bin/bash
# round 5, call 16: where the 2^N - 1 grid's time goes (kernel stats at n = 1023), rocFFT back-end for its length-1024 transforms
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
QP_LINES=40 bash scripts/quick_prof.sh r05/c16_n1023 --n 1023 > /dev/null 2>&1
HPS_POISSON_BACKEND=rocfft python bench.py --cpu-slices 0 --n 1023 --inflight 1 > $O/c16_n1023_rocfft.json 2>> $O/c16.err
python bench.py --cpu-slices 0 --n 511 --inflight 1 > $O/c16_n511.json 2>> $O/c16.err
python bench.py --cpu-slices 0 --n 512 --inflight 1 > $O/c16_n512.json 2>> $O/c16.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c16_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), {k: (round(v, 4) if v else v) for k, v in d["phase_ms_per_slice"].items()}, d["vcycles_per_slice"])
PY
cut -c1-100,100-200 $O/c16_n1023_kstats.txt | head -40
