import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes, time, sys
from membrane_solver_amd import _lib as L, meshgen
f=int(sys.argv[1]) if len(sys.argv)>1 else 320
P,T=meshgen.icosphere(f); P=meshgen.smooth_displace(P,0.05)
lib=L.lib()
st=(ctypes.c_int64*8)()
perm=np.zeros(len(P),dtype=np.int32)
Pc=np.ascontiguousarray(P); Tc=np.ascontiguousarray(T.astype(np.int32))
t=time.time()
rc=lib.ms_plan_tiling(len(P),len(T),Pc.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),Tc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),256,1,st,perm.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
print(rc,list(st),"%.2fs"%(time.time()-t))
print("instances/nf %.4f  avg facets/tile %.1f max %d max_halo %d"%(st[1]/len(T), st[1]/st[0], st[3], st[2]))
m=(ctypes.c_double*4)()
rc=lib.ms_plan_tiling_conflicts(len(P),len(T),Pc.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),Tc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),256,m)
print("bank model",[round(x,3) for x in m])
