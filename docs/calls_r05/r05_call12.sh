#!/bin/bash
# round 5, call 12: pool of 3 re-checked; the new bench line (per_kernel, phase window); fused push+deposit under stages in flight; whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/c12_bench_steps20.json 2>> $O/c12.err
python bench.py --cpu-slices 0 --steps 2048 --inflight 3 --ring-self --edge ipc > $O/c12_ringself_3st_ipc.json 2>> $O/c12.err
timeout 600 python bench.py --gpus 2 --same-device --steps 20 --warmup 5 --cpu-slices 0 > $O/c12_2ranks_same_device.json 2>> $O/c12.err
timeout 600 python bench.py --gpus 2 --same-device --steps 1024 --cpu-slices 0 > $O/c12_2ranks_same_device_wholebox.json 2>> $O/c12.err
for L in 1 2 3; do
  python bench.py --cpu-slices 0 --inflight $L > $O/c12_sep_inflight$L.json 2>> $O/c12.err
  python bench.py --cpu-slices 0 --inflight $L --fuse > $O/c12_fuse_inflight$L.json 2>> $O/c12.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/c12_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f.split("/")[-1], round(d.get("value", d.get("slices_per_s")), 1), d.get("value_steps_in_flight"), d.get("ranks_seen"), (d.get("in_flight") or {}).get("error"))
PY
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/c12_gpu_suite.txt; cat $O/c12_gpu_suite.txt
