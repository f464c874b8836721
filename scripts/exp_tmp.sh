python -m pytest tests -m gpu -x -q -k "advance_plasma or slice_by_slice or ioniz or laser_wake" 2>&1 | grep -E "passed|failed" | tail -2
pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_slice']; print(round(d['value'],1), d.get('value_steps_in_flight'), round(p['advance_plasma'],4))"; }
python bench.py --cpu-slices 0 | pj
python bench.py --cpu-slices 0 | pj
python bench.py --config5 --steps 2048 | pj
