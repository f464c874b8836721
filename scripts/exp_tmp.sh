pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['steps'], d['steps_in_flight'], d['value_steps_in_flight'], d['in_flight'])"; }
python bench.py --cpu-slices 0 | pj
python bench.py --cpu-slices 0 --steps 20 --warmup 5 | pj
python bench.py --cpu-slices 0 --steps 20 --warmup 5 --inflight 2 | pj
python bench.py --cpu-slices 0 --steps 20 --warmup 5 --inflight 4 | pj
python bench.py --config2 --steps 512 | pj
