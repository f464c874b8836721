pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d.get('value_steps_in_flight'))"; }
for i in 1 2 3; do HPS_DRIVE_READY=0 python bench.py --cpu-slices 0 --steps 20 --warmup 5 | pj; python bench.py --cpu-slices 0 --steps 20 --warmup 5 | pj; done
HPS_DRIVE_READY=0 python bench.py --cpu-slices 0 | pj
python bench.py --cpu-slices 0 | pj
