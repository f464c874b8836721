#!/bin/bash
# round 5, call 29: two processes on one device under rocprofv3: do their kernels overlap on the device?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
rm -rf /tmp/prof_2p
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_2p/%pid% -o kt -- python $R/bench.py --gpus 2 --same-device --steps 1024 --cpu-slices 0 --one-stage-per-rank > $O/c29_bench.json 2> $O/c29.err)
find /tmp/prof_2p -name "*kernel_trace.csv" | head; 
python scripts/two_process_overlap.py /tmp/prof_2p > $O/c29_two_process_overlap.txt 2>&1; cat $O/c29_two_process_overlap.txt; tail -2 $O/c29_bench.json | cut -c1-200
