pj() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d.get('value_steps_in_flight'))"; }
python bench.py --cpu-slices 0 | pj
GPU_MAX_HW_QUEUES=8 python bench.py --cpu-slices 0 | pj
GPU_MAX_HW_QUEUES=2 python bench.py --cpu-slices 0 | pj
python bench.py --cpu-slices 0 --inflight 4 --steps 4096 | pj
GPU_MAX_HW_QUEUES=8 python bench.py --cpu-slices 0 --inflight 4 --steps 4096 | pj
python bench.py --cpu-slices 0 --inflight 2 --steps 2048 | pj
