import sys, time, numpy as np, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from axtrack_amd import params, hotpath as hp
from axtrack_amd.detections import transition_cost_table, _arc_cost_int_vec
from oracle import oracle as orc
def network(rep_t=4, rep_s=2, F_lim=None):
    d = np.load('/root/repo/tests/data/c3_dets.npz')
    F0 = len(d['count'])
    F = F0 * rep_t if F_lim is None else F_lim
    X=[];Y=[];C=[]
    for t in range(F):
        tt = t % F0; n = d['count'][tt]
        xs=[];ys=[];cs=[]
        for a in range(rep_s):
            for b in range(rep_s):
                xs.append(d['x'][tt,:n].astype(np.int64)+512*a); ys.append(d['y'][tt,:n].astype(np.int64)+512*b); cs.append(d['conf'][tt,:n])
        X.append(np.concatenate(xs)); Y.append(np.concatenate(ys)); C.append(np.concatenate(cs))
    cnt = np.array([len(x) for x in X])
    table, dmax = transition_cost_table(params.DEPLOYED)
    offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    tails, heads, gaps, lens = [], [], [], []
    for t in range(F):
        for g in (1, 2):
            tb = t + g
            if tb >= F: continue
            dx = np.abs(X[t][:, None] - X[tb][None]); dy = np.abs(Y[t][:, None] - Y[tb][None])
            D = dx + dy + 1
            i, j = np.nonzero((D <= dmax[g - 1]) & (dx * dx + dy * dy < 250000))
            tails.append(offs[t] + i); heads.append(offs[tb] + j); gaps.append(np.full(len(i), g)); lens.append(D[i, j])
    a, b, g, L = (np.concatenate(v) for v in (tails, heads, gaps, lens))
    order = np.lexsort((b, g, a)); a, b, g, L = a[order], b[order], g[order], L[order]
    cost = _arc_cost_int_vec(np.where(g == 1, table[0][L], table[1][L]), 3, a, b)
    n = int(offs[-1]); row_ptr = np.zeros(n + 1, np.int64); row_ptr[1:] = np.cumsum(np.bincount(a, minlength=n))
    conf = np.concatenate(C).astype(np.float64)
    obs = orc.observation_cost(orc.cap_conf(conf)); k = np.arange(n)
    return (_arc_cost_int_vec(obs, 2, k, 0), _arc_cost_int_vec(np.full(n, 2.0), 0, k, 0), _arc_cost_int_vec(np.full(n, 2.0), 1, k, 0), row_ptr, b.astype(np.int32), cost, offs)
if __name__ == '__main__':
    net = network(int(sys.argv[1]), int(sys.argv[2]))
    print(len(net[0]), 'detections', len(net[4]), 'arcs', flush=True)
    np.savez('/tmp/c4net.npz', *net); sys.exit(0)
    for _ in range(2):
        t=time.perf_counter(); res = hp.mcf_solve(*net[:6], 5, 450*4); print('solve', time.perf_counter()-t, res[2], res[3], flush=True)
