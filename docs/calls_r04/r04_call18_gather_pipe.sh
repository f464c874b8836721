#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
run () { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab18_$name.json 2>> $O/ab18.err
  python - "$name" "$O/ab18_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:20s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f}  ", {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run base_a --
run gp1_a HPS_LIB=$R/hipace_amd/csrc/libhpslice_gp1.so --
run gp2_a HPS_LIB=$R/hipace_amd/csrc/libhpslice_gp2.so --
run base_b --
run gp1_b HPS_LIB=$R/hipace_amd/csrc/libhpslice_gp1.so --
run gp2_b HPS_LIB=$R/hipace_amd/csrc/libhpslice_gp2.so --
