#!/bin/bash
# round 5, call 23: counter traffic of the node-centred multigrid's kernels (n = 1023) next to the cell-centred ones (n = 1024)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
for n in 1023 1024; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -o pmc -- python $R/bench.py --cpu-slices 0 --inflight 1 --n $n --steps 24 --warmup 4 --start-slice 600 --phase-window 0 > /dev/null 2>&1)
  cp $(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1) /tmp/pmc_$C.csv
done
python scripts/pmc_traffic.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv > $O/c23_pmc_n$n.csv
done
grep -E "k_smooth|k_restrict|k_lower" $O/c23_pmc_n1023.csv | cut -c1-60,100-200
echo ----
grep -E "k_smooth|k_restrict|k_lower" $O/c23_pmc_n1024.csv | cut -c1-60,100-200
