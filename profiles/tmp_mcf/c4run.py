import sys, time, numpy as np
sys.path.insert(0, '.')
from axtrack_amd import hotpath as hp
z = np.load('/tmp/c4net.npz'); net = [z[f'arr_{i}'] for i in range(7)]
for _ in range(3):
    t = time.perf_counter(); res = hp.mcf_solve(*net[:6], 5, 1800); print('c4 solve %.3f s' % (time.perf_counter() - t), res[2], res[3], flush=True)
