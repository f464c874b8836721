#!/bin/bash
# the whole GPU suite, smoke() and the driver's bench line on the shipped build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04/suite_final.log 2>&1
grep -E "passed|failed" gpurun_out/r04/suite_final.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r04/suite_final.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/final_steps20.json 2> gpurun_out/r04/final.err
python bench.py > gpurun_out/r04/final_plain.json 2>> gpurun_out/r04/final.err
python - <<'PY'
import json
for f in ("final_steps20","final_plain"):
    d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"],1), d.get("value_steps_in_flight"), d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"] if "cpu_baseline" in d else None)
PY
