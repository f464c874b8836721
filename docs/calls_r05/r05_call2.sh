#!/bin/bash
# round 5, call 2: the ipc edge between processes on one device: parity tests, C++ host, bench with 2 ranks on one device
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_ring_processes_gpu.py -x -q 2>&1 | tail -40 > gpurun_out/r05/c2_ipc_tests.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "closing_edge or ring_driver_on_rccl or cpp or rccl_ring" 2>&1 | tail -15 > gpurun_out/r05/c2_selfring_tests.txt
timeout 600 python bench.py --gpus 2 --same-device --steps 20 --warmup 5 --cpu-slices 0 > gpurun_out/r05/c2_bench_2ranks_same_device.json 2> gpurun_out/r05/c2_bench_2ranks_same_device.err
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-slices 0 > gpurun_out/r05/c2_bench_1rank.json 2> gpurun_out/r05/c2_bench_1rank.err
tail -5 gpurun_out/r05/c2_ipc_tests.txt gpurun_out/r05/c2_selfring_tests.txt
tail -3 gpurun_out/r05/c2_bench_2ranks_same_device.err
python - <<'PY'
import json
for f in ("c2_bench_2ranks_same_device", "c2_bench_1rank"):
    try:
        d = json.loads(open(f"gpurun_out/r05/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d.get("value_steps_in_flight"), d.get("ranks_seen"), d.get("ring_edge"), d.get("ring"))
    except Exception as e:
        print(f, "no line:", e)
PY
