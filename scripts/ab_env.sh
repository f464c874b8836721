#!/bin/bash
# A/B of one HPS_* switch on one box: bash scripts/ab_env.sh VAR "<bench args>" [repeats]; prints value (and value_steps_in_flight) per run
VAR=$1; ARGS=$2; N=${3:-2}
for i in $(seq $N); do for v in 1 0; do
  env $VAR=$v python bench.py --cpu-slices 0 $ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(j['value'],1), j.get('value_steps_in_flight'))"
done; done
