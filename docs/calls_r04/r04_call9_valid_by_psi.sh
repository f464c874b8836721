#!/bin/bash
# round 4, GPU call 9: the push without its idcpu read (HPS_VALID_BY_PSI, default on in this build) -- parity (schedule test, push operator tests,
# whole boxes of the BASELINE configs) and A/B with one and three stages in flight, alone and with the two other byte cuts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "schedules_do_not_change or advance_plasma or engine_slice_by_slice or tiled_operators" > $O/vbp_tests.log 2>&1
tail -3 $O/vbp_tests.log
timeout 900 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -x > $O/vbp_fullsize.log 2>&1
tail -3 $O/vbp_fullsize.log
run () {   # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab9_$name.json 2>> $O/ab9.err
  python - "$name" "$O/ab9_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:28s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ",
          {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run vbp0_a HPS_VALID_BY_PSI=0 --
run vbp1_a --
run vbp0_b HPS_VALID_BY_PSI=0 --
run vbp1_b --
run all_cuts_a HPS_VALID_BY_W=1 HPS_POISSON_BLOCKED=1 --
run all_cuts_b HPS_VALID_BY_W=1 HPS_POISSON_BLOCKED=1 --
run all_cuts_L4 HPS_VALID_BY_W=1 HPS_POISSON_BLOCKED=1 -- --inflight 4
run all_cuts_c3 HPS_VALID_BY_W=1 HPS_POISSON_BLOCKED=1 -- --n 512
run base_c3 HPS_VALID_BY_PSI=0 -- --n 512
tail -3 $O/ab9.err
