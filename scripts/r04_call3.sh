#!/bin/bash
# round 4, GPU call 3: per-kernel profiles of the blocked / transposing Poisson solve, deposit prototypes (flush / atomics apart)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
QP_LINES=24 bash scripts/quick_prof.sh r04/p_blocked
QP_LINES=24 HPS_POISSON_BLOCKED=0 bash scripts/quick_prof.sh r04/p_unblocked
timeout 600 python scripts/deposit_variants.py > gpurun_out/r04/deposit_variants2.txt 2> gpurun_out/r04/deposit_variants2.err
cat gpurun_out/r04/deposit_variants2.txt
export HPS_FULLSIZE_REPORT=gpurun_out/r04/fullsize
timeout 1500 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -s -k "config2 or config5" > gpurun_out/r04/fullsize_tests3.log 2>&1
grep -E "^config|passed|failed|^\.config|^sconfig|^Fconfig|Error" gpurun_out/r04/fullsize_tests3.log | head
