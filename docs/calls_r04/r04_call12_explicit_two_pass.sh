#!/bin/bash
# round 4, GPU call 12: the explicit deposition in two passes (no per-cell branches): parity + A/B against -DHPS_EXPL_TWO_PASS=0
# sub-cycles; engine slice by slice; goldens), whole boxes, A/B against -DHPS_PUSH_SPLIT_GATHER=0 and a 4-waves-per-SIMD build (128 VGPRs, 148 B scratch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "explicit or deposit or advance_plasma or engine_slice_by_slice or tiled_operators or golden or laser or ioniz" > $O/ex_tests.log 2>&1
grep -E "passed|failed" $O/ex_tests.log | tail -2
timeout 900 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -x > $O/ex_fullsize.log 2>&1
grep -E "passed|failed" $O/ex_fullsize.log | tail -2
run () {   # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab12_$name.json 2>> $O/ab12.err
  python - "$name" "$O/ab12_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:28s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ",
          {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
A0=HPS_LIB=$R/hipace_amd/csrc/libhpslice_ex0.so
run ex0_a $A0 --
run ex1_a --
run ex0_b $A0 --
run ex1_b --
run ex0_c5 $A0 -- --config5
run ex1_c5 -- --config5
run ex0_c3 $A0 -- --n 512
run ex1_c3 -- --n 512
tail -3 $O/ab12.err
