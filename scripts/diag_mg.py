import sys, time; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from hipace_amd import api, _lib
L=_lib.lib()
L.hps_mg_debug_stamps.argtypes=[C.c_void_p]
n=int(sys.argv[1]) if len(sys.argv)>1 else 1024
g=2
f=api.Fields(n,n,g,5)
f.t[2:4,g:-g,g:-g]=torch.randn((2,n,n),dtype=torch.float64,device='cuda')
f.t[4]=0.5+torch.rand((n+2*g,n+2*g),dtype=torch.float64,device='cuda')
mg=api.MultiGrid(n,n,16/n,16/n)
st=(C.c_longlong*48)()
L.hps_mg_debug_stamps(st)
it,rn=mg.solve1(f,0,2,4)
torch.cuda.synchronize()
t=time.time(); 
for _ in range(10):
    f.t[0:2]=0
    it,rn=mg.solve1(f,0,2,4)
torch.cuda.synchronize(); dt=(time.time()-t)/10
L.hps_mg_debug_stamps(st)
v=list(st)
print('iters',it,'ms/solve',dt*1e3)
print('k_smooth (last launch, block 0) deltas:', [v[i+1]-v[i] for i in range(5)])
print('k_lower_v deltas:', [v[i+1]-v[i] for i in range(8,14)])
print('wave-part down stamps (level: ticks since stamp 11):', {l: v[16+l]-v[11] for l in range(1,9) if v[16+l]})
print('wave-part up stamps:', {l: v[32+l]-v[11] for l in range(1,9) if v[32+l]}, 'end', v[12]-v[11])
