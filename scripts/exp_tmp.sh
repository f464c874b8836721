run() { echo "== $*"; env "$@" python bench.py --cpu-slices 0 --steps 1024 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_slice']; print(round(d['value'],1), round(p['advance_plasma'],4))"; }
for t in w3 w3p w4p w4; do run HPS_LIB=$PWD/hipace_amd/csrc/libhpslice_$t.so; done
