#!/bin/bash
# round 4, GPU call 22: the shipped build (r04i = r04g + Poisson leftover scheme for small factors, 32-bit offsets in the multigrid's views, DPP stopping
# rule): whole GPU suite, smoke(), the profile set (kernel stats, PMC FETCH/WRITE + SQ), the bench lines of every config
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/suite_i.log 2>&1
grep -E "passed|failed" $O/suite_i.log | tail -2; grep -E "^FAILED|^ERROR" $O/suite_i.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/collect_profiles.sh r04/r04i > $O/collect_i.log 2>&1; tail -22 $O/collect_i.log | cut -c1-200
python bench.py --config2 > $O/r04i_config2.json 2>> $O/i.err
python bench.py --n 512 > $O/r04i_config3.json 2>> $O/i.err
python bench.py --config5 > $O/r04i_config5_fft.json 2>> $O/i.err
python bench.py --config5 --laser-solver multigrid > $O/r04i_config5_mg.json 2>> $O/i.err
python - <<'PY'
import json
for f in ("r04i_bench_plain", "r04i_bench_steps20", "r04i_config2", "r04i_config3", "r04i_config5_fft", "r04i_config5_mg"):
    try:
        d = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), d.get("value_steps_in_flight"), d["roofline"]["frac"] if d.get("roofline") else None)
    except Exception as e:
        print(f, "FAILED", e)
PY
