#!/bin/bash
# round 4, GPU call 16: the shipped build (Poisson leftover scheme for small factors, psi_half guards): whole GPU suite, smoke(), the driver's line, config 3 / config 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_h.log 2>&1
grep -E "passed|failed" $O/suite_h.log | tail -2; grep -E "^FAILED|^ERROR" $O/suite_h.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/r04h_bench_steps20.json 2>> $O/h.err
python bench.py > $O/r04h_bench_plain.json 2>> $O/h.err
python bench.py --n 512 > $O/r04h_config3.json 2>> $O/h.err
python bench.py --config2 > $O/r04h_config2.json 2>> $O/h.err
python - <<'PY'
import json
for f in ("r04h_bench_steps20", "r04h_bench_plain", "r04h_config3", "r04h_config2"):
    try:
        d = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), d.get("value_steps_in_flight"), d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("whole_box_elsewhere"))
    except Exception as e:
        print(f, "FAILED", e)
PY
