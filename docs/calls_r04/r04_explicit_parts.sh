#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
( python scripts/explicit_parts.py
for v in NO_FLUSH NO_READS NO_ATOMICS NO_AR NONE; do HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_ex_$v.so python scripts/explicit_parts.py; done ) > gpurun_out/r04/explicit_parts.txt 2> gpurun_out/r04/explicit_parts.err
cat gpurun_out/r04/explicit_parts.txt; tail -3 gpurun_out/r04/explicit_parts.err
