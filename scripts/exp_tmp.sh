python -m pytest tests -m gpu -x -q -k "slice_by_slice or golden or reproduces or schedules or fused or ioniz or laser or density or mobile or orders" 2>&1 | grep -E "passed|failed|Error" | tail -3
pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_slice']; print(round(d['value'],1), d.get('value_steps_in_flight'), {k: round(v,4) for k,v in p.items() if v})"; }
python bench.py --cpu-slices 0 | pj
