#!/bin/bash
# round 5, call 3: three stages per rank, closing edge in-process / through the ring (ipc, rccl): rates, stage phases, host timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
export GPU_MAX_HW_QUEUES=8 NCCL_MAX_P2P_NCHANNELS=2
for st in 2 3; do
for edge in none ipc rccl; do
  arg=""; [ $edge != none ] && arg="--ring-self $edge"
  python scripts/inflight_run.py --stages $st --boxes 2 $arg > $O/c3_rate_${st}st_$edge.json 2>> $O/c3.err
done; done
for edge in none ipc; do
  arg=""; [ $edge != none ] && arg="--ring-self $edge"
  rm -rf /tmp/prof_if
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_if -o kt -- python $R/scripts/inflight_run.py --stages 3 --boxes 1 $arg > $O/c3_trace_3st_$edge.json 2>/dev/null)
  CSV=$(find /tmp/prof_if -name "*kernel_trace.csv" | head -1)
  python scripts/overlap_analysis.py $CSV > $O/c3_overlap_3st_$edge.txt
  python scripts/stage_phase.py $CSV > $O/c3_phase_3st_$edge.txt
  python scripts/gap_analysis.py $CSV > $O/c3_gaps_3st_$edge.txt 2>&1
  HPS_DRIVE_TRACE=$O/c3_hosttrace_3st_$edge.txt python scripts/inflight_run.py --stages 3 --boxes 1 $arg > $O/c3_hosttrace_3st_$edge.json 2>> $O/c3.err
done
cat $O/c3_rate_*.json; cat $O/c3_phase_3st_*.txt; head -12 $O/c3_overlap_3st_*.txt
