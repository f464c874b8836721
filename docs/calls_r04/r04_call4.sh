#!/bin/bash
# round 4, GPU call 4: whole suite with the device-controlled predictor-corrector loop, config 2 A/B, headline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04/suite4.log 2>&1
grep -E "passed|failed|error" gpurun_out/r04/suite4.log | tail -3
grep -E "^FAILED|Error|assert" gpurun_out/r04/suite4.log | head -20
for s in 0 1; do
  HPS_PC_SPECULATE=$s python bench.py --config2 --inflight 1 > gpurun_out/r04/c2_spec$s.json 2>> gpurun_out/r04/c2.err
done
python bench.py --config2 --inflight 3 --steps 1536 > gpurun_out/r04/c2_spec1_inflight3.json 2>> gpurun_out/r04/c2.err
python bench.py --config2 --inflight 2 --steps 1536 > gpurun_out/r04/c2_spec1_inflight2.json 2>> gpurun_out/r04/c2.err
python bench.py > gpurun_out/r04/bench4_plain.json 2>> gpurun_out/r04/bench4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/c2_*.json"))+["gpurun_out/r04/bench4_plain.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), d.get("value_steps_in_flight"), d.get("pc_iterations_per_slice"), {k:(round(v,4) if v else v) for k,v in d["phase_ms_per_slice"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 gpurun_out/r04/c2.err
