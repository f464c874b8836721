#!/bin/bash
# round 4, GPU call 19: the switches that were measured "no gain" in round 3, again on this round's build, with one and three stages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
run () { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" > $O/ab19_$name.json 2>> $O/ab19.err
  python - "$name" "$O/ab19_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:20s} one stage {d['value']:8.1f}   in flight {d.get('value_steps_in_flight') or 0:8.1f} (L={d.get('steps_in_flight')})  ", {k: round(1e3 * v, 1) for k, v in d["phase_ms_per_slice"].items() if v})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run base_a --
run post_fold HPS_MG_POST_FOLD=1 --
run aux_stream HPS_AUX_STREAM=1 --
run drive_ready HPS_DRIVE_READY=1 --
run vbp HPS_VALID_BY_PSI=1 --
run nt0_lazy0 HPS_LAZY_SHIFT=0 --
run fallback128 HPS_SORT_FALLBACK_DIV=128 --
run fallback512 HPS_SORT_FALLBACK_DIV=512 --
run base_b --
