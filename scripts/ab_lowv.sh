#!/bin/bash
# A/B of the node-centred lower V (k_lower_v) at 1023^2: small levels by wave 0 alone (HPS_MG_LOWV_WAVE), one component per workgroup, threads
for v in "0 1 1024" "1 1 1024" "1 1 512" "1 1 256" "1 0 1024" "1089 1 1024"; do set -- $v
  echo "== wave=$1 split=$2 threads=$3"
  HPS_MG_LOWV_WAVE=$1 HPS_MG_LOWV_SPLIT=$2 HPS_MG_LOWV_THREADS=$3 python -m pytest tests/test_gpu_parity.py -x -q -k "multigrid_solve1 and (1023 or 511 or 255 or 63 or 31)" 2>&1 | tail -1
  HPS_MG_LOWV_WAVE=$1 HPS_MG_LOWV_SPLIT=$2 HPS_MG_LOWV_THREADS=$3 python bench.py --n 1023 --cpu-slices 0 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], round(d['roofline']['per_kernel']['mg_solve1']['us_per_slice'],1))"
done
