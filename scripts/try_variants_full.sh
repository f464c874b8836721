for so in "$@"; do
  cp hipace_amd/csrc/$so hipace_amd/csrc/libhpslice.so
  echo "== $so"
  python bench.py --cpu-slices 0 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(v,4) for k,v in d['phase_ms_per_slice'].items()}, d['particle_sorts'], d['halo_fallbacks'])"
done
