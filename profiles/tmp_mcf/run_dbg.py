import sys, ctypes, time, numpy as np, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/profiles')
import mcf_timing as m
rep = int(sys.argv[1]) if len(sys.argv)>1 else 1
lib = ctypes.CDLL(sys.argv[2] if len(sys.argv)>2 else os.path.dirname(os.path.abspath(__file__))+'/libmcfdbg.so')
net = m.network(rep)
obs,en,ex,rp,col,cost = [np.ascontiguousarray(a) for a in net]
n=len(obs)
nxt=np.empty(n,np.int32); tr=np.empty(n,np.int32); nt=ctypes.c_int(0); tot=ctypes.c_int64(0)
lib.axt_mcf_solve.argtypes=[ctypes.c_int]+[ctypes.c_void_p]*6+[ctypes.c_int,ctypes.c_int]+[ctypes.c_void_p]*2+[ctypes.POINTER(ctypes.c_int),ctypes.POINTER(ctypes.c_int64)]
for _ in range(int(os.environ.get('REPS','1'))):
    t=time.perf_counter()
    rc=lib.axt_mcf_solve(n,obs.ctypes.data,en.ctypes.data,ex.ctypes.data,rp.ctypes.data,col.ctypes.data,cost.ctypes.data,5,450*rep,nxt.ctypes.data,tr.ctypes.data,ctypes.byref(nt),ctypes.byref(tot))
    print('rc',rc,'ms',1e3*(time.perf_counter()-t),'tracks',nt.value,'cost',tot.value)
