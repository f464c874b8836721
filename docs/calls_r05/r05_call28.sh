#!/bin/bash
# round 5, call 28: LDS bank swizzle of the power-of-two DST kernel's working set (N = 1024): parity, A/B at n = 1023
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "poisson or head_slices" 2>&1 | tail -2
for i in 1 2; do
  python bench.py --cpu-slices 0 --n 1023 --inflight 1 > $O/c28_swz_$i.json 2>> $O/c28.err
  HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_p2_noswz.so python bench.py --cpu-slices 0 --n 1023 --inflight 1 > $O/c28_noswz_$i.json 2>> $O/c28.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c28_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), round(d["phase_ms_per_slice"]["poisson"], 4))
PY
