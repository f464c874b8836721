#!/bin/bash
# round 4, GPU call 1: suite + whole-box parity + baseline bench + halo / non-temporal A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export HPS_FULLSIZE_REPORT=gpurun_out/r04/fullsize
timeout 1500 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -s > gpurun_out/r04/fullsize_tests.log 2>&1
unset HPS_FULLSIZE_REPORT
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_fullsize_boxes.py > gpurun_out/r04/suite.log 2>&1
tail -3 gpurun_out/r04/suite.log
python bench.py > gpurun_out/r04/bench_plain.json 2> gpurun_out/r04/bench_plain.err
for v in "" _h5 _h4 _nt; do
  HPS_LIB=$PWD/hipace_amd/csrc/libhpslice$v.so python bench.py --inflight 1 --cpu-slices 0 > gpurun_out/r04/ab$v.json 2>> gpurun_out/r04/ab.err
done
HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_h4.so python bench.py --cpu-slices 0 --steps 3072 > gpurun_out/r04/ab_h4_inflight.json 2>> gpurun_out/r04/ab.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/ab*.json"))+["gpurun_out/r04/bench_plain.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), d.get("value_steps_in_flight"), {k:(round(v,4) if v else v) for k,v in d["phase_ms_per_slice"].items()}, d["particle_sorts"], d["halo_fallbacks"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -15 gpurun_out/r04/fullsize_tests.log
