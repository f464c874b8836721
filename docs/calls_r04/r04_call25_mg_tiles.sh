#!/bin/bash
# round 4, GPU call 25: the multigrid's tile choices per level again, now that threads keep their block's values in registers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run () { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --cpu-slices 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name'.ljust(34), round(d['value'],1), round(d['value_steps_in_flight'] or 0,1), 'mg', round(1e3*d['phase_ms_per_slice']['mg_solve1'],1))"
}
run base_a --
run init_huge HPS_MG_INIT_HUGE=1 --
run l2_mid HPS_MG_SMALL_CELLS=20000 HPS_MG_MID_CELLS=70000 --
run l1_l2_mid HPS_MG_SMALL_CELLS=20000 HPS_MG_MID_CELLS=300000 --
run l1_mid_only HPS_MG_MID_CELLS=300000 --
run l1_small HPS_MG_SMALL_CELLS=300000 --
run base_b --
