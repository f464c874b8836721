#!/bin/bash
# round 4, GPU call 23: the gated push posts the Bx/By solve's norms to the host itself (no k_post_norms launch between the last V-cycle and the push):
# parity (schedules, engine, whole boxes, local pipelines), A/B against HPS_POST_IN_PUSH=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "schedules or engine_slice_by_slice or golden or local_pipeline or smoke" > $O/pip_tests.log 2>&1
grep -E "passed|failed" $O/pip_tests.log | tail -2
timeout 600 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -x > $O/pip_fullsize.log 2>&1
grep -E "passed|failed" $O/pip_fullsize.log | tail -2
run () { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --cpu-slices 0 "$@" > $O/ab23_$name.json 2>> $O/ab23.err
  python - "$name" "$O/ab23_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:20s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ", {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run pip0_a HPS_POST_IN_PUSH=0 --
run pip1_a --
run pip0_b HPS_POST_IN_PUSH=0 --
run pip1_b --
run pip0_c3 HPS_POST_IN_PUSH=0 -- --n 512
run pip1_c3 -- --n 512
