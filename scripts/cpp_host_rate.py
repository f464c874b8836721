"""Rate of the C++ host (examples/ring_host) on the headline deck: three boxes through the RCCL ring, no Python in the loop."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipace_amd import api, decks
deck = decks.synthetic(1024, 1024, 2)
e = api.SliceEngine(deck, tile_size=16, sort_period=128)
open("/tmp/deck.bin", "wb").write(bytes(e._dk))
del e
env = dict(os.environ, RING_HOST_NO_DIAG="1")
r = subprocess.run([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "ring_host"), "/tmp/deck.bin", "3", "16", "128"],
                   capture_output=True, text=True, env=env)
print(r.stderr[-600:])
print(r.stdout.splitlines()[-1])
