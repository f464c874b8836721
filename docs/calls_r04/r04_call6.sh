#!/bin/bash
# round 4, GPU call 6: config 2 with the deeper dense product and the fence-free post; multigrid level-0 tile 64 x 96; config 5 whole boxes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predictor or poisson or schedules or bench or reference_checksums" > $O/tests6.log 2>&1; tail -3 $O/tests6.log
export HPS_FULLSIZE_REPORT=$O/fullsize
timeout 900 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -s -k "config2" > $O/fullsize_tests6.log 2>&1
unset HPS_FULLSIZE_REPORT
grep -E "^config|passed|failed|^\.config|^sconfig|^Fconfig|Error" $O/fullsize_tests6.log | head
for s in 0 1; do
  HPS_PC_SPECULATE=$s python bench.py --config2 --inflight 1 > $O/c2b_spec$s.json 2>> $O/c2b.err
done
python bench.py --config2 --inflight 3 --steps 1536 > $O/c2b_spec1_inflight3.json 2>> $O/c2b.err
bash scripts/collect_mfma.sh r04/r04d > $O/collect_mfma.log 2>&1
for v in "" _mg96 _mg96k; do
  HPS_LIB=$R/hipace_amd/csrc/libhpslice$v.so python bench.py --inflight 1 --cpu-slices 0 > $O/mg_tile$v.json 2>> $O/mg_tile.err
done
HPS_LIB=$R/hipace_amd/csrc/libhpslice_mg96.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multigrid_solve1 or baseline_blowout" > $O/mg96_tests.log 2>&1; tail -2 $O/mg96_tests.log
python bench.py --config5 --cpu-slices 0 > $O/c5_fft_whole.json 2>> $O/c5.err
python bench.py --config5 --laser-solver multigrid --cpu-slices 0 > $O/c5_mg_whole.json 2>> $O/c5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/c2b_*.json"))+sorted(glob.glob("gpurun_out/r04/mg_tile*.json"))+sorted(glob.glob("gpurun_out/r04/c5_*_whole.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), d.get("value_steps_in_flight"), d.get("pc_iterations_per_slice"), d.get("vcycles_per_slice"), d.get("laser_vcycles_per_slice"), {k:(round(v,4) if v else v) for k,v in d["phase_ms_per_slice"].items()}, d.get("timed_slices"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/c2b.err $O/c5.err $O/mg_tile.err
