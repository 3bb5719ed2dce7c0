import sys, ctypes, time, numpy as np, os
z=np.load(os.environ.get('C4NET','/tmp/c4net.npz')); net=[np.ascontiguousarray(z[f'arr_{i}']) for i in range(7)]
obs,en,ex,rp,col,cost,offs = net
lib = ctypes.CDLL(''+os.path.dirname(os.path.abspath(__file__))+'/libmcfdbg.so')
n=len(obs)
nxt=np.empty(n,np.int32); tr=np.empty(n,np.int32); nt=ctypes.c_int(0); tot=ctypes.c_int64(0)
lib.axt_mcf_solve.argtypes=[ctypes.c_int]+[ctypes.c_void_p]*6+[ctypes.c_int,ctypes.c_int]+[ctypes.c_void_p]*2+[ctypes.POINTER(ctypes.c_int),ctypes.POINTER(ctypes.c_int64)]
t=time.perf_counter()
rc=lib.axt_mcf_solve(n,obs.ctypes.data,en.ctypes.data,ex.ctypes.data,rp.ctypes.data,col.ctypes.data,cost.ctypes.data,5,1800,nxt.ctypes.data,tr.ctypes.data,ctypes.byref(nt),ctypes.byref(tot))
print('rc',rc,'s',time.perf_counter()-t,'tracks',nt.value,'cost',tot.value)
