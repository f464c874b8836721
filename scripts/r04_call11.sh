#!/bin/bash
# round 4, GPU call 11: final profile set of the shipped build (r04f), config 2 / config 5 lines, one ring diagnostic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
bash scripts/collect_profiles.sh r04/r04f > $O/collect_r04f.log 2>&1; tail -5 $O/collect_r04f.log | cut -c1-200
python bench.py --config2 > $O/r04f_config2.json 2>> $O/c11.err
bash scripts/collect_mfma.sh r04/r04f > $O/collect_mfma_f.log 2>&1
python bench.py --config5 --cpu-slices 0 > $O/r04f_config5_fft.json 2>> $O/c11.err
python bench.py --config5 --laser-solver multigrid --cpu-slices 0 > $O/r04f_config5_mg.json 2>> $O/c11.err
python bench.py --n 512 --cpu-slices 0 > $O/r04f_config3.json 2>> $O/c11.err
GPU_MAX_HW_QUEUES=8 NCCL_MAX_P2P_NCHANNELS=2 HPS_RING_SELF_COPY=1 python bench.py --cpu-slices 0 --steps 3072 --inflight 3 --ring-self > $O/ringself_3boxes_selfcopy.json 2>> $O/c11.err
python - <<'PY'
import json
for f in ("r04f_bench_plain","r04f_bench_steps20","r04f_bench_under_rocprof","r04f_config2","r04f_config5_fft","r04f_config5_mg","r04f_config3","ringself_3boxes_selfcopy"):
    try:
        d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1]); print(f, round(d["value"],1), d.get("value_steps_in_flight"), d["roofline"]["frac"], d.get("pc_iterations_per_slice"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/c11.err
