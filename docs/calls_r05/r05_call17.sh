#!/bin/bash
# round 5, call 17: SI config-5 boxes against their fixtures, the ring tests with the probe message, whole-box reports of all configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O $O/fullsize
timeout 900 python -m pytest tests/test_ring_processes_gpu.py -x -q 2>&1 | tail -5 > $O/c17_ring_tests.txt
HPS_FULLSIZE_REPORT=$O/fullsize timeout 2400 python -m pytest tests/test_fullsize_boxes.py -q -s 2>&1 | grep -E "passed|failed|worst|FAILED|Error|error" | tail -30 > $O/c17_fullsize.txt
cat $O/c17_ring_tests.txt $O/c17_fullsize.txt; ls $O/fullsize
