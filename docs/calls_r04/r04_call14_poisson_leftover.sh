#!/bin/bash
# round 4, GPU call 14: the 41-point (1024^2) / 19-point (512^2) DFT stage with the items that do not fill a wave packed into one wave of their own
# (sym_stage LEFTOVER scheme): Poisson parity, A/B against -DHPS_SYM_LEFTOVER=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "poisson or engine_slice_by_slice or golden or full_size or schedules" > $O/pl_tests.log 2>&1
grep -E "passed|failed" $O/pl_tests.log | tail -2
timeout 900 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -x > $O/pl_fullsize.log 2>&1
grep -E "passed|failed" $O/pl_fullsize.log | tail -2
run () {   # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab14_$name.json 2>> $O/ab14.err
  python - "$name" "$O/ab14_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:28s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ",
          {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
A0=HPS_LIB=$R/hipace_amd/csrc/libhpslice_pl0.so
run pl0_a $A0 --
run pl1_a --
run pl0_b $A0 --
run pl1_b --
run pl0_nb $A0 HPS_POISSON_BLOCKED=0 --
run pl1_nb HPS_POISSON_BLOCKED=0 --
run pl0_c3 $A0 -- --n 512
run pl1_c3 -- --n 512
run pl0_c5 $A0 -- --config5
run pl1_c5 -- --config5
tail -3 $O/ab14.err
