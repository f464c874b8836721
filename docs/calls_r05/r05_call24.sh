#!/bin/bash
# round 5, call 24: the node-centred fused 8-sweep pass (39 spilled registers): prolongation through LDS (base), smaller tile, more registers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "multigrid_solve1 or head_slices" 2>&1 | tail -2
for v in base mg_h6432 mg_nw3 mg_nw2; do
  lib=$PWD/hipace_amd/csrc/libhpslice_$v.so; [ $v = base ] && lib=$PWD/hipace_amd/csrc/libhpslice.so
  HPS_LIB=$lib python bench.py --cpu-slices 0 --n 1023 --inflight 1 > $O/c24_$v.json 2>> $O/c24.err
done
HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_mg_h6432.so python bench.py --cpu-slices 0 --inflight 1 > $O/c24_n1024_mg_h6432.json 2>> $O/c24.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c24_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), round(d["phase_ms_per_slice"]["mg_solve1"], 4), d["vcycles_per_slice"])
PY
