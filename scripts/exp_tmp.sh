pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['phase_ms_per_slice']['poisson'],4))"; }
for s in 0 1 2 3 4; do echo stagger $s; HPS_DST_STAGGER=$s python bench.py --cpu-slices 0 --inflight 1 | pj; done
