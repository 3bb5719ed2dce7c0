#!/usr/bin/env python3
"""Times axt_mcf_solve (host C++) on the flow network of the C3 bench timelapse (19 340 detections, 553 k arcs,
built from tests/data/c3_dets.npz exactly as tests/test_host_logic.py does). Runs without a GPU.
  python profiles/mcf_timing.py [repeat]        AXT_MCF_DEBUG=1 prints the solver's search statistics"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from axtrack_amd import params, hotpath as hp
from axtrack_amd.detections import transition_cost_table, _arc_cost_int_vec
from oracle import oracle as orc


def network(rep=1):
    d = np.load(os.path.join(ROOT, 'tests', 'data', 'c3_dets.npz'))
    cnt = np.tile(d['count'], rep)
    F = len(cnt)
    X = [d['x'][t % len(d['count']), :cnt[t]].astype(np.int64) for t in range(F)]
    Y = [d['y'][t % len(d['count']), :cnt[t]].astype(np.int64) for t in range(F)]
    table, dmax = transition_cost_table(params.DEPLOYED)
    offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    tails, heads, gaps, lens = [], [], [], []
    for t in range(F):
        for g in (1, 2):
            tb = t + g
            if tb >= F:
                continue
            dx = np.abs(X[t][:, None] - X[tb][None]); dy = np.abs(Y[t][:, None] - Y[tb][None])
            D = dx + dy + 1
            i, j = np.nonzero((D <= dmax[g - 1]) & (dx * dx + dy * dy < 250000))
            tails.append(offs[t] + i); heads.append(offs[tb] + j); gaps.append(np.full(len(i), g)); lens.append(D[i, j])
    a, b, g, L = (np.concatenate(v) for v in (tails, heads, gaps, lens))
    order = np.lexsort((b, g, a))
    a, b, g, L = a[order], b[order], g[order], L[order]
    cost = _arc_cost_int_vec(np.where(g == 1, table[0][L], table[1][L]), 3, a, b)
    n = int(offs[-1])
    row_ptr = np.zeros(n + 1, np.int64)
    row_ptr[1:] = np.cumsum(np.bincount(a, minlength=n))
    conf = np.concatenate([d['conf'][t % len(d['count']), :cnt[t]] for t in range(F)]).astype(np.float64)
    obs = orc.observation_cost(orc.cap_conf(conf))
    k = np.arange(n)
    return (_arc_cost_int_vec(obs, 2, k, 0), _arc_cost_int_vec(np.full(n, 2.0), 0, k, 0),
            _arc_cost_int_vec(np.full(n, 2.0), 1, k, 0), row_ptr, b.astype(np.int32), cost)


if __name__ == '__main__':
    rep = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    net = network(rep)
    print(f'{len(net[0])} detections, {len(net[4])} arcs')
    for _ in range(3):
        t = time.perf_counter()
        res = hp.mcf_solve(*net, 5, 450 * rep)
        print(f'solve {1e3 * (time.perf_counter() - t):.1f} ms  tracks {res[2]}  cost {res[3]}')
