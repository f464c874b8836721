#!/bin/bash
# round 5, call 32: the final build (r05c): whole GPU suite, smoke(), profile set (kernel stats, PMC FETCH/WRITE + SQ), bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/suite_c.log 2>&1
grep -E "passed|failed" $O/suite_c.log | tail -2; grep -E "^FAILED|^ERROR" $O/suite_c.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/collect_profiles.sh r05/r05c > $O/collect_c.log 2>&1; tail -16 $O/collect_c.log | cut -c1-200
python bench.py --n 1023 --cpu-slices 0 > $O/r05c_n1023.json 2>> $O/c.err
python bench.py --n 511 --cpu-slices 0 > $O/r05c_n511.json 2>> $O/c.err
python bench.py --config2 > $O/r05c_config2.json 2>> $O/c.err
python bench.py --n 512 > $O/r05c_config3.json 2>> $O/c.err
python bench.py --config5 > $O/r05c_config5_fft.json 2>> $O/c.err
python bench.py --config5 --laser-solver multigrid > $O/r05c_config5_mg.json 2>> $O/c.err
timeout 600 python bench.py --gpus 2 --same-device --steps 20 --warmup 5 > $O/r05c_2ranks_same_device_steps20.json 2>> $O/c.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/r05c_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"), d["roofline"]["frac"] if d.get("roofline") else None, d.get("ranks_seen"))
    except Exception as e:
        print(f, "FAILED", e)
PY
