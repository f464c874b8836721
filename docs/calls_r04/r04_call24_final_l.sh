#!/bin/bash
# round 4, GPU call 24: the shipped build at the end of round 4 (r04l = r04i + the push that posts the norms + block rows in the smoother): whole GPU suite, smoke(), profile set, bench lines
# rule): whole GPU suite, smoke(), the profile set (kernel stats, PMC FETCH/WRITE + SQ), the bench lines of every config
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_l.log 2>&1
grep -E "passed|failed" $O/suite_l.log | tail -2; grep -E "^FAILED|^ERROR" $O/suite_l.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/collect_profiles.sh r04/r04l > $O/collect_l.log 2>&1; tail -22 $O/collect_l.log | cut -c1-200
python bench.py --config2 > $O/r04l_config2.json 2>> $O/l.err
python bench.py --n 512 > $O/r04l_config3.json 2>> $O/l.err
python bench.py --config5 > $O/r04l_config5_fft.json 2>> $O/l.err
python bench.py --config5 --laser-solver multigrid > $O/r04l_config5_mg.json 2>> $O/l.err
python - <<'PY'
import json
for f in ("r04l_bench_plain", "r04l_bench_steps20", "r04l_config2", "r04l_config3", "r04l_config5_fft", "r04l_config5_mg"):
    try:
        d = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), d.get("value_steps_in_flight"), d["roofline"]["frac"] if d.get("roofline") else None)
    except Exception as e:
        print(f, "FAILED", e)
PY
