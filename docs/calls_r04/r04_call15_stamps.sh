#!/bin/bash
# round 4, GPU call 15: shader-clock stamps of workgroup 0 of the Poisson row kernel (libhpslice_stamps.so): what a pass's time is made of
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
HPS_LIB=$R/hipace_amd/csrc/libhpslice_stamps.so HPS_POISSON_BLOCKED=0 python scripts/diag_poisson.py 2>&1 | tail -8 | tee $O/poisson_stamps.txt
HPS_LIB=$R/hipace_amd/csrc/libhpslice_stamps.so python scripts/diag_poisson.py 2>&1 | tail -3 | tee -a $O/poisson_stamps.txt
