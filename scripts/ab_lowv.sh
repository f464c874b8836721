#!/bin/bash
# A/B of the node-centred lower V (k_lower_v): one component per workgroup and threads per workgroup, at 1023^2
for v in "0 1024" "1 1024" "1 512" "1 256" "0 512" "0 256"; do set -- $v
  echo "== split=$1 threads=$2"
  HPS_MG_LOWV_SPLIT=$1 HPS_MG_LOWV_THREADS=$2 python -m pytest tests/test_gpu_parity.py -x -q -k "multigrid_solve1 and (1023 or 511 or 255 or 63 or 31)" 2>&1 | tail -1
  HPS_MG_LOWV_SPLIT=$1 HPS_MG_LOWV_THREADS=$2 python bench.py --n 1023 --cpu-slices 0 --inflight 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], round(d['roofline']['per_kernel']['mg_solve1']['us_per_slice'],1))"
done
