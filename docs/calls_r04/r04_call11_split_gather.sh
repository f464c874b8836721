#!/bin/bash
# round 4, GPU call 11: the push's gather with the plain fields on their 3 x 3 cells only (52 instead of 80 LDS reads): parity (all orders, boundaries,
# sub-cycles; engine slice by slice; goldens), whole boxes, A/B against -DHPS_PUSH_SPLIT_GATHER=0 and a 4-waves-per-SIMD build (128 VGPRs, 148 B scratch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "advance_plasma or engine_slice_by_slice or tiled_operators or golden or laser or ioniz" > $O/sg_tests.log 2>&1
grep -E "passed|failed" $O/sg_tests.log | tail -2
timeout 900 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -x > $O/sg_fullsize.log 2>&1
grep -E "passed|failed" $O/sg_fullsize.log | tail -2
run () {   # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab11_$name.json 2>> $O/ab11.err
  python - "$name" "$O/ab11_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:28s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ",
          {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
A0=HPS_LIB=$R/hipace_amd/csrc/libhpslice_sg0.so
W4=HPS_LIB=$R/hipace_amd/csrc/libhpslice_w4.so
run sg0_a $A0 --
run sg1_a --
run w4_a $W4 --
run sg0_b $A0 --
run sg1_b --
run sg0_c5 $A0 -- --config5
run sg1_c5 -- --config5
run sg0_c3 $A0 -- --n 512
run sg1_c3 -- --n 512
tail -3 $O/ab11.err
