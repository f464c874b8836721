#!/bin/bash
# round 4, GPU call 21: the multigrid's field views indexed by 32-bit byte offsets (HPS_MG_OFF32): parity + A/B against -DHPS_MG_OFF32=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "multigrid or mg or engine_slice_by_slice or golden or full_size or laser" > $O/mg32_tests.log 2>&1
grep -E "passed|failed" $O/mg32_tests.log | tail -2
timeout 900 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -x > $O/mg32_fullsize.log 2>&1
grep -E "passed|failed" $O/mg32_fullsize.log | tail -2
run () { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab21_$name.json 2>> $O/ab21.err
  python - "$name" "$O/ab21_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:20s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ", {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
A0=HPS_LIB=$R/hipace_amd/csrc/libhpslice_mg0.so
run mg0_a $A0 --
run mg1_a --
run mg0_b $A0 --
run mg1_b --
run mg0_c3 $A0 -- --n 512
run mg1_c3 -- --n 512
run mg0_c5mg $A0 -- --config5 --laser-solver multigrid
run mg1_c5mg -- --config5 --laser-solver multigrid
run mg0_c2 $A0 -- --config2
run mg1_c2 -- --config2
