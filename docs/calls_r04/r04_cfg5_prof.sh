#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/prof_cfg5.sh r04/r04f_c5fft > /dev/null 2>&1
bash scripts/prof_cfg5.sh r04/r04f_c5mg --laser-solver multigrid > /dev/null 2>&1
head -24 gpurun_out/r04/r04f_c5fft_kstats.txt | cut -c1-64,100-215
