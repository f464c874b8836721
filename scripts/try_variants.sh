#!/bin/bash
# usage: try_variants.sh so1 so2 ... : run the bench with each library variant
for so in "$@"; do
  cp hipace_amd/csrc/$so hipace_amd/csrc/libhpslice.so
  echo "== $so"
  python bench.py --steps 192 --warmup 32 --cpu-slices 0 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v,4) for k,v in d['phase_ms_per_slice'].items()})"
done
