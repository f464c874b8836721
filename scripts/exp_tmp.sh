python -m pytest tests -m gpu -x -q -k "multigrid2 or laser or ioniz" 2>&1 | grep -E "passed|failed" | tail -2
pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_slice']; print(round(d['value'],1), {k: round(v,4) for k,v in p.items() if v}, d.get('laser_vcycles_per_slice'))"; }
python bench.py --config5 --steps 2048 | pj
python bench.py --config5 --steps 2048 --laser-solver multigrid | pj
python bench.py --config5 --steps 2048 --laser-solver multigrid --no-ionization | pj
