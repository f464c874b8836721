#!/bin/bash
# round 5, call 20: rows and threads per workgroup of the power-of-two DST kernel (n = 1023): Poisson phase per slice
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
for v in base t1_n128 t1_n256 t2_n128 t2_n512 t4_n512 t4_n256; do
  lib=$PWD/hipace_amd/csrc/libhpslice_p2_$v.so; [ $v = base ] && lib=$PWD/hipace_amd/csrc/libhpslice.so
  HPS_LIB=$lib python bench.py --cpu-slices 0 --n 1023 --inflight 1 --steps 256 --warmup 16 --start-slice 600 --profile-stride 4 --phase-window 0 > $O/c20_$v.json 2>> $O/c20.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c20_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), d["phase_ms_per_slice"].get("poisson"), d["phase_ms_per_slice"].get("deposit_current"))
PY
tail -3 $O/c20.err
