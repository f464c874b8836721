python -m pytest tests -m gpu -x -q -k "multigrid or slice_by_slice or golden" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_stamps.so python scripts/diag_mg.py 1024 2>&1 | tail -4
pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_slice']; print(round(d['value'],1), d.get('value_steps_in_flight'), round(p['mg_solve1'],4))"; }
python bench.py --cpu-slices 0 | pj
