#!/bin/bash
# round 4, GPU call 2: blocked Poisson parity + A/B, halo / non-temporal A/B, deposit prototypes, config 4 whole box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "poisson or reference_checksums or slice_by_slice or baseline_blowout or schedules" > gpurun_out/r04/poisson_tests.log 2>&1
tail -3 gpurun_out/r04/poisson_tests.log
export HPS_FULLSIZE_REPORT=gpurun_out/r04/fullsize
timeout 1500 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -s -k "config4 or config2" > gpurun_out/r04/fullsize_tests2.log 2>&1
unset HPS_FULLSIZE_REPORT
grep -E "^config|passed|failed|^\.config|^sconfig|^Fconfig" gpurun_out/r04/fullsize_tests2.log | head
HPS_POISSON_BLOCKED=0 python bench.py --inflight 1 --cpu-slices 0 > gpurun_out/r04/ab2_unblocked.json 2>> gpurun_out/r04/ab2.err
python bench.py --inflight 1 --cpu-slices 0 > gpurun_out/r04/ab2_blocked.json 2>> gpurun_out/r04/ab2.err
for v in _h5 _h4 _nt; do
  HPS_LIB=$PWD/hipace_amd/csrc/libhpslice$v.so python bench.py --inflight 1 --cpu-slices 0 > gpurun_out/r04/ab2$v.json 2>> gpurun_out/r04/ab2.err
done
HPS_POISSON_BLOCKED=0 python bench.py --inflight 1 --cpu-slices 0 > gpurun_out/r04/ab2_unblocked_b.json 2>> gpurun_out/r04/ab2.err
python bench.py --inflight 1 --cpu-slices 0 > gpurun_out/r04/ab2_blocked_b.json 2>> gpurun_out/r04/ab2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/ab2*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), {k:(round(v,4) if v else v) for k,v in d["phase_ms_per_slice"].items()}, d["particle_sorts"], d["halo_fallbacks"])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 600 python scripts/deposit_variants.py > gpurun_out/r04/deposit_variants.txt 2> gpurun_out/r04/deposit_variants.err
cat gpurun_out/r04/deposit_variants.txt; tail -5 gpurun_out/r04/deposit_variants.err
