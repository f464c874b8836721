#!/bin/bash
# A/B of k_tridiag_y's shape (rows per thread x columns per workgroup) on batches of three solves
for v in "16 16" "16 8" "16 4" "32 8" "32 4"; do set -- $v; echo "== M=$1 COLS=$2"; HPS_TRI_M=$1 HPS_TRI_COLS=$2 python scripts/diag_poisson_tri.py 64x64 1024x1024 1023x1023 512x512 2>&1 | grep "tridiag=1"; done
