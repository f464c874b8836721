#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04; timeout 600 python scripts/deposit_variants.py > gpurun_out/r04/deposit_variants3.txt 2> gpurun_out/r04/deposit_variants3.err
cat gpurun_out/r04/deposit_variants3.txt; tail -3 gpurun_out/r04/deposit_variants3.err
