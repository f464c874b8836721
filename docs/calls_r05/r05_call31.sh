#!/bin/bash
# round 5, call 31: the push's order-2 weights computed once (nodal_weights2_w3): parity, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "advance_plasma or tiled_operators or head_slices" 2>&1 | tail -2
for i in 1 2; do
  python bench.py --cpu-slices 0 --inflight 3 > $O/c31_w3_$i.json 2>> $O/c31.err
  HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_now3.so python bench.py --cpu-slices 0 --inflight 3 > $O/c31_now3_$i.json 2>> $O/c31.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c31_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), round(d["value_steps_in_flight"], 1), round(d["phase_ms_per_slice"]["advance_plasma"], 4))
PY
