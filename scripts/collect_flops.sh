#!/bin/bash
# fp64 instruction counts of the particle kernels (rocprofv3 --pmc, its own pass with --kernel-trace only): per-wave counts of the
# SQ's ADD / MUL / FMA / TRANS fp64 instructions -> flops per launch = 64 lanes x (ADD + MUL + 2 FMA + TRANS).
# Run on the GPU box from the repo root:  bash scripts/collect_flops.sh r06  -> gpurun_out/<tag>_pmc_fp64_per_kernel.csv
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*" | sort -u > $OUT/${TAG}_avail_valu_counters.txt
rm -rf /tmp/prof_fl
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU --output-format csv -d /tmp/prof_fl -o pmc -- python $GRAFT_REPO_ROOT/bench.py --cpu-slices 0 --inflight 1 --steps 24 --warmup 4 --start-slice 36 > /dev/null 2>$OUT/${TAG}_flops_err.txt
F=$(find /tmp/prof_fl -name "*counter_collection.csv" | head -1)
python - "$F" > $OUT/${TAG}_pmc_fp64_per_kernel.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r['Kernel_Name'][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES': cnt[k] += 1
names = sorted({c for v in acc.values() for c in v})
print('kernel,launches,' + ','.join(names) + ',fp64_flops_per_launch')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0)):
    n = max(cnt[k], 1)
    fl = 64.0*(v.get('SQ_INSTS_VALU_ADD_F64', 0) + v.get('SQ_INSTS_VALU_MUL_F64', 0) + 2*v.get('SQ_INSTS_VALU_FMA_F64', 0) + v.get('SQ_INSTS_VALU_TRANS_F64', 0))/n
    print('"%s",%d,' % (k, n) + ','.join('%.5g' % (v.get(c, 0)/n) for c in names) + ',%.5g' % fl)
PY
head -8 $OUT/${TAG}_pmc_fp64_per_kernel.csv | cut -c1-260
