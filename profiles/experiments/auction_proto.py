#!/usr/bin/env python3
"""Round counts of the epsilon-scaling forward/reverse auction prototype (auction_proto.cpp) on the tracker's networks.
    python profiles/experiments/auction_proto.py static [rep] | moving c3|c4 | c4 [frames]     (THETA=8, EPS_SHIFT=0 by env)"""
import ctypes, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from axtrack_amd import hotpath as hp
from tests import helpers
so = '/tmp/auction_proto.so'
subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'profiles/experiments/auction_proto.cpp')])
lib = ctypes.CDLL(so)
lib.auction_solve.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
kind = sys.argv[1] if len(sys.argv) > 1 else 'static'
if kind == 'moving':
    F, size, alive = {'c3': (252, 512, 90), 'c4': (1020, 1024, 380)}[sys.argv[2]]
    net = helpers.moving_network(F, size, alive)[:6]; max_flow = 100000
elif kind == 'c4':
    d = np.load('/tmp/c4_dets.npz'); F = int(sys.argv[2]) if len(sys.argv) > 2 else len(d['count'])
    net = helpers.open_grid_network(d['count'][:F], d['x'][:F].astype(np.int64), d['y'][:F].astype(np.int64), d['conf'][:F], 1024, 1024)[:6]; max_flow = 100000
else:
    rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    d = np.load(os.path.join(ROOT, 'tests', 'data', 'c3_dets.npz'))
    t = np.arange(len(d['count']) * rep) % len(d['count'])
    net = helpers.open_grid_network(d['count'][t], d['x'][t], d['y'][t], d['conf'][t], 512, 512)[:6]; max_flow = 450 * rep
obs, en, ex, row_ptr, col, cost = [np.ascontiguousarray(a) for a in net]
n = len(obs)
print(n, 'detections', len(col), 'arcs')
t = time.perf_counter(); ref = hp.mcf_solve(obs, en, ex, row_ptr, col, cost, 0, 10 ** 9); print('host solver %.1f ms, tracks %d' % (1e3 * (time.perf_counter() - t), ref[2]))
out = np.empty(n, np.int32); price = np.empty(n, np.int64); stats = np.zeros(4, np.int64)
t = time.perf_counter()
ph = lib.auction_solve(n, obs.ctypes.data, en.ctypes.data, ex.ctypes.data, row_ptr.ctypes.data, col.ctypes.data, cost.ctypes.data,
                       out.ctypes.data, price.ctypes.data, int(os.environ.get('THETA', 8)), 1, ctypes.c_int64(int(os.environ.get('EPS_SHIFT', 0))), stats.ctypes.data)
print('auction %.1f s: phases %d, forward rounds %d, reverse rounds %d, bids %d, offers %d' % (time.perf_counter() - t, ph, *stats))
nxt, track = ref[0], ref[1]
exp = np.where(track < 0, np.arange(n), np.where(nxt < 0, n + np.arange(n), nxt))
print('rows that differ from the exact solution:', int((exp != out).sum()))
