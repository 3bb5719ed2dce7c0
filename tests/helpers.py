"""Shared test helpers (CPU side): build the arc list the GPU kernel would, from oracle matrices."""
import numpy as np

from oracle import oracle as orc


def split(counts, *arrs):
    offs = np.concatenate([[0], np.cumsum(counts)])
    return [tuple(a[offs[i]:offs[i + 1]] for a in arrs) for i in range(len(counts))]


def golden_dets(g):
    return split(g['counts'], g['conf'], g['x'], g['y'])


def csr_arcs_from_oracle(dets, H, W, P=orc.DEFAULTS, mask=None, name='synth'):
    """CSR (row_ptr, col, length, gap, cost_int) by tail detection, rows sorted by (gap, b) --
    the layout of axt_build_arcs -- computed with the oracle's path matrices and costs."""
    counts = [len(d[0]) for d in dets]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    gaps = P['MCF_MAX_NUM_MISSES'] + 1
    rows = [[] for _ in range(int(offs[-1]))]
    for t in range(len(dets)):
        for g in range(1, gaps + 1):
            tb = t + g
            if tb >= len(dets) or counts[t] == 0 or counts[tb] == 0:
                continue
            D = orc.path_matrix(dets[t], dets[tb], H, W, mask)
            c = orc.transition_cost(D, g, P['MCF_MISS_RATE'])
            for i, j in zip(*np.nonzero(c < P['MCF_EDGE_COST_THR'])):
                a, b = int(offs[t] + i), int(offs[tb] + j)
                rows[a].append((g, b, int(D[i, j]), orc.arc_cost_int(c[i, j], 3, a, b)))
    row_ptr, col, length, gap, cost = [0], [], [], [], []
    for r in rows:
        for g, b, d, ci in sorted(r):
            col.append(b); length.append(d); gap.append(g); cost.append(ci)
        row_ptr.append(len(col))
    return (np.array(row_ptr, np.int64), np.array(col, np.int32), np.array(length, np.int16),
            np.array(gap, np.uint8), np.array(cost, np.int64), offs)


def node_costs_from_oracle(dets, P=orc.DEFAULTS):
    conf = np.concatenate([np.asarray(d[0], np.float32) for d in dets]).astype(np.float64)
    obs = orc.observation_cost(orc.cap_conf(conf, P['MCF_CONF_CAPPING_METHOD']), P['MCF_MAX_CONF_COST'])
    n = len(conf)
    ee = float(P['MCF_ENTRY_EXIT_COST'])
    obs_i = np.array([orc.arc_cost_int(obs[k], 2, k, 0) for k in range(n)], np.int64)
    en_i = np.array([orc.arc_cost_int(ee, 0, k, 0) for k in range(n)], np.int64)
    ex_i = np.array([orc.arc_cost_int(ee, 1, k, 0) for k in range(n)], np.int64)
    return obs_i, en_i, ex_i, obs


def tracks_from_next(nxt, track, offs):
    """product output -> list of trajectories [(frame, idx), ...] ordered by track id."""
    n = len(nxt)
    frame_of = np.searchsorted(offs, np.arange(n), side='right') - 1
    out = {}
    for k in range(n):
        if track[k] >= 0:
            out.setdefault(int(track[k]), []).append((int(frame_of[k]), int(k - offs[frame_of[k]])))
    return [sorted(out[i]) for i in sorted(out)]
