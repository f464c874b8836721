#!/bin/bash
# round 5, call 15: deposition without the IEEE division (A/B), the 2^N - 1 grid (bench line + head-slices parity), 
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for i in 1 2; do
  python bench.py --cpu-slices 0 --inflight 3 > $O/c15_base_$i.json 2>> $O/c15.err
  HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_deprcp.so python bench.py --cpu-slices 0 --inflight 3 > $O/c15_deprcp_$i.json 2>> $O/c15.err
done
python bench.py --cpu-slices 0 --n 1023 > $O/c15_n1023.json 2>> $O/c15.err
python bench.py --cpu-slices 0 --n 1023 --steps 20 --warmup 5 > $O/c15_n1023_steps20.json 2>> $O/c15.err
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "head_slices" 2>&1 | tail -3 > $O/c15_head_slices_test.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c15_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"), {k: (round(v, 4) if v else v) for k, v in d["phase_ms_per_slice"].items()})
PY
cat $O/c15_head_slices_test.txt
