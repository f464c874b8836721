python -m pytest tests -m gpu -x -q -k "poisson or golden or slice_by_slice" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_slice']; print(round(d['value'],1), d.get('value_steps_in_flight'), round(p['poisson'],4))"; }
python bench.py --cpu-slices 0 | pj
HPS_POISSON_Y2=0 python bench.py --cpu-slices 0 | pj
