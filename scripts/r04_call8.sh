#!/bin/bash
# round 4, GPU call 8: valid-by-weight as kernel variants (A/B), the whole suite, the driver's bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for v in 1 0 1 0; do
  HPS_VALID_BY_W=$v python bench.py --inflight 1 --cpu-slices 0 > $O/vbwt$v.json 2>> $O/vbwt.err
  python - <<PY
import json
d=json.loads(open("$O/vbwt$v.json").read().strip().splitlines()[-1]); print("valid_by_w=$v", round(d["value"],1), {k:(round(v,4) if v else v) for k,v in d["phase_ms_per_slice"].items()})
PY
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite8.log 2>&1
grep -E "passed|failed" $O/suite8.log | tail -2; grep -E "^FAILED|^E  " $O/suite8.log | head
python bench.py --steps 20 --warmup 5 > $O/bench8_steps20.json 2>> $O/bench8.err
python bench.py > $O/bench8_plain.json 2>> $O/bench8.err
python - <<'PY'
import json
for f in ("bench8_steps20","bench8_plain"):
    d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"],1), d.get("value_steps_in_flight"), d["roofline"]["frac"], d["roofline"]["slice"], (d.get("in_flight") or {}).get("roofline"))
PY
