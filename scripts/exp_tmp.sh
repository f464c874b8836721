python -m pytest tests -m gpu -x -q -k "slice_by_slice or golden or reproduces or schedules or fused or full_size or orders or config" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_slice']; print(round(d['value'],1), d.get('value_steps_in_flight'), {k: round(v,4) for k,v in p.items() if v})"; }
python bench.py --cpu-slices 0 | pj
HPS_FUSE_SOURCES=0 python bench.py --cpu-slices 0 | pj
