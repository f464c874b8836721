#!/bin/bash
# round 4, GPU call 12: depth of the dense product's operand prefetch (config 2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for rep in 1 2; do
for v in "" _dd1 _dd2 _dd4; do
  HPS_LIB=$R/hipace_amd/csrc/libhpslice$v.so python bench.py --config2 --inflight 1 > $O/dd$v.$rep.json 2>> $O/dd.err
  python - <<PY
import json
d=json.loads(open("$O/dd$v.$rep.json").read().strip().splitlines()[-1]); ph=d["phase_ms_per_slice"]
print("depth variant '$v'", round(d["value"],1), "its/slice", round(d["pc_iterations_per_slice"],3), "loop us/iteration", round(1e3*ph["mg_solve1"]/d["pc_iterations_per_slice"],1))
PY
done; done
