#!/bin/bash
# round 5, call 27: the final build (r05b = r05a + node-centred multigrid changes, ring clean-ups): whole GPU suite, smoke(), the driver's command lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/suite_b.log 2>&1
grep -E "passed|failed" $O/suite_b.log | tail -2; grep -E "^FAILED|^ERROR" $O/suite_b.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05b_bench_steps20.json 2>> $O/b.err
python bench.py > $O/r05b_bench_plain.json 2>> $O/b.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 --same-device > $O/r05b_N2_same_device_steps20.json 2>> $O/b.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/r05b_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"], 1), d.get("value_steps_in_flight"), d["roofline"]["frac"], d.get("ranks_seen"), d["roofline"].get("dominant_by_time"))
    except Exception as e:
        print(f, "FAILED", e)
PY
