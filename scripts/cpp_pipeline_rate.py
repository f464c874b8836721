"""examples/pipeline_host (C++ host, one thread, several stages per rank) on the headline workload: slices/s for 1-4 stages.
usage (GPU box, repo root): python scripts/cpp_pipeline_rate.py"""
import os, subprocess, sys, tempfile
sys.path.insert(0, '.')
from hipace_amd import api, decks
import __graft_entry__
__graft_entry__.build_cpp_host()
deck = decks.synthetic(1024, 1024, 2)
eng = api.SliceEngine(deck)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "deck.bin")
open(path, "wb").write(bytes(eng._dk))
del eng
env = dict(os.environ, PIPELINE_HOST_NO_DIAG="1")
for stages in (1, 2, 3, 4):
    out = subprocess.run(["examples/pipeline_host", path, str(2 * stages + 1), "0", "1", tmp, str(stages)], capture_output=True, text=True, env=env, timeout=900)
    line = [l for l in out.stderr.splitlines() if "slices/s" in l]
    print(stages, "stage(s):", line[-1] if line else out.stderr[-300:])
