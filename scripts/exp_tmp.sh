run() { echo "== $*"; env "$@" python bench.py --cpu-slices 0 --steps 2048 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['phase_ms_per_slice']['mg_solve1'],4))"; }
run A=1
run HPS_MG_INIT_HUGE=0
run A=1
run HPS_MG_INIT_HUGE=0
