#!/bin/bash
# round 4, GPU call 10: the push's six sub-steps with the field products hoisted and the zeta derivative written out by hand (42 instead of 62 fp64
# instructions per sub-step): parity tests, whole boxes, A/B against the same build with -DHPS_PUSH_ALGEBRA=0 (libhpslice_alg0.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "advance_plasma or engine_slice_by_slice or tiled_operators or golden or laser or ioniz" > $O/alg_tests.log 2>&1
tail -3 $O/alg_tests.log
timeout 900 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -x > $O/alg_fullsize.log 2>&1
tail -3 $O/alg_fullsize.log
run () {   # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab10_$name.json 2>> $O/ab10.err
  python - "$name" "$O/ab10_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:28s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ",
          {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
A0=HPS_LIB=$R/hipace_amd/csrc/libhpslice_alg0.so
run alg0_a $A0 --
run alg1_a --
run alg0_b $A0 --
run alg1_b --
run alg0_c5 $A0 -- --config5
run alg1_c5 -- --config5
run alg0_c5mg $A0 -- --config5 --laser-solver multigrid
run alg1_c5mg -- --config5 --laser-solver multigrid
run alg0_c3 $A0 -- --n 512
run alg1_c3 -- --n 512
run alg0_c2 $A0 -- --config2
run alg1_c2 -- --config2
tail -3 $O/ab10.err
