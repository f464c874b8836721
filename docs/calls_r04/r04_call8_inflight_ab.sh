#!/bin/bash
# round 4, GPU call 8: byte cuts and compute-unit masks measured WITH SEVERAL STAGES IN FLIGHT (the earlier A/Bs were --inflight 1,
# where no particle kernel but the push is bound by its bytes; three stages move 4.07 TB/s in aggregate)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
run () {   # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab8_$name.json 2>> $O/ab8.err
  python - "$name" "$O/ab8_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    inf = d.get("in_flight") or {}
    print(f"{sys.argv[1]:28s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ",
          {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run base_a -- 
run valid_by_w HPS_VALID_BY_W=1 --
run poisson_blocked HPS_POISSON_BLOCKED=1 --
run vbw_blocked HPS_VALID_BY_W=1 HPS_POISSON_BLOCKED=1 --
run fuse3 BENCH_FUSE_INFLIGHT=1 -- --fuse
run fuse2 BENCH_FUSE_INFLIGHT=1 -- --fuse --inflight 2
run base_L2 -- --inflight 2
run cu_all_L3 HPS_CU_MASKS=0-255 --
run cu128_L1 HPS_CU_MASKS=0-127 -- --inflight 1
run cu192_L1 HPS_CU_MASKS=0-191 -- --inflight 1
run cu64_L1 HPS_CU_MASKS=0-63 -- --inflight 1
run cu_halves_L2 HPS_CU_MASKS=0-127,128-255 -- --inflight 2
run cu_thirds_L3 HPS_CU_MASKS=0-87,88-175,176-255 --
run cu_overlap_L2 HPS_CU_MASKS=0-191,64-255 -- --inflight 2
run cu_overlap_L3 HPS_CU_MASKS=0-159,48-207,96-255 --
run base_b --
tail -5 $O/ab8.err
