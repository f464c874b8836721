#!/bin/bash
# round 4, GPU call 7: predictor-corrector noise floor (config 2 whole box strict), valid-by-weight depositions, host posts without the system fence
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
export HPS_FULLSIZE_REPORT=$O/fullsize
timeout 900 python -m pytest tests/test_fullsize_boxes.py -m gpu -q -s > $O/fullsize_tests7.log 2>&1
unset HPS_FULLSIZE_REPORT
grep -E "^config|passed|failed|^\.config|^sconfig|^Fconfig|Error" $O/fullsize_tests7.log | head -12
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_fullsize_boxes.py > $O/suite7.log 2>&1
grep -E "passed|failed" $O/suite7.log | tail -2; grep -E "^FAILED|^E  " $O/suite7.log | head
for v in 1 0 1 0; do
  HPS_VALID_BY_W=$v python bench.py --inflight 1 --cpu-slices 0 > $O/vbw$v.json 2>> $O/vbw.err
  python - <<PY
import json
d=json.loads(open("$O/vbw$v.json").read().strip().splitlines()[-1]); print("valid_by_w=$v", round(d["value"],1), {k:(round(v,4) if v else v) for k,v in d["phase_ms_per_slice"].items()})
PY
done
HPS_LIB=$R/hipace_amd/csrc/libhpslice_sysfence.so python bench.py --inflight 1 --cpu-slices 0 > $O/sysfence.json 2>> $O/vbw.err
python bench.py --config2 > $O/c2c.json 2>> $O/c2c.err
HPS_PC_NOISE_FLOOR=0 python bench.py --config2 > $O/c2c_nofloor.json 2>> $O/c2c.err
python - <<'PY'
import json
for f in ("sysfence","c2c","c2c_nofloor"):
    d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"],1), d.get("value_steps_in_flight"), d.get("pc_iterations_per_slice"), {k:(round(v,4) if v else v) for k,v in d["phase_ms_per_slice"].items()})
PY
bash scripts/quick_prof.sh r04/r04e > /dev/null 2>&1; head -30 $O/r04e_kstats.txt | cut -c1-60,100-215
