#!/bin/bash
# round 5, call 21: push with the deposition as its tail (k_advance_tiled<.., DEP>): parity of the fused schedule, rates with 1 and 3 stages, kernel time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused" 2>&1 | tail -3 > $O/c21_tests.txt; cat $O/c21_tests.txt
for L in 1 3; do
  python bench.py --cpu-slices 0 --inflight $L > $O/c21_sep_$L.json 2>> $O/c21.err
  python bench.py --cpu-slices 0 --inflight $L --fuse > $O/c21_fuse_$L.json 2>> $O/c21.err
  HPS_FUSED_KERNEL=old python bench.py --cpu-slices 0 --inflight $L --fuse > $O/c21_fuseold_$L.json 2>> $O/c21.err
done
QP_LINES=8 bash scripts/quick_prof.sh r05/c21_fuse --fuse > /dev/null 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c21_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"))
PY
cut -c1-100,100-200 $O/c21_fuse_kstats.txt | head -8
