"""Shared test helpers (CPU side): build the arc list the GPU kernel would, from oracle matrices."""
import os

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


def open_grid_network(count, x, y, conf, H, W, P=None):
    """The flow network of a set of detections on an all-ones mask, built on the CPU exactly as axt_build_arcs + the
    host side of assign_ids build it: (obs, entry, exit i64 [n], row_ptr i64 [n+1], col i32, cost i64, offs, dets)."""
    from axtrack_amd import params
    from axtrack_amd.detections import transition_cost_table, _arc_cost_int_vec
    P = P or params.DEPLOYED
    cnt = np.asarray(count)
    F = len(cnt)
    X = [x[t, :cnt[t]].astype(np.int64) for t in range(F)]
    Y = [y[t, :cnt[t]].astype(np.int64) for t in range(F)]
    table, dmax = transition_cost_table(P)
    offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    tails, heads, gaps, lens = [], [], [], []
    for t in range(F):
        inb_a = (X[t] >= 0) & (X[t] < W) & (Y[t] >= 0) & (Y[t] < H)
        for g in range(1, len(dmax) + 1):
            tb = t + g
            if tb >= F:
                continue
            dx = np.abs(X[t][:, None] - X[tb][None]); dy = np.abs(Y[t][:, None] - Y[tb][None])
            inb = inb_a[:, None] & ((X[tb] >= 0) & (X[tb] < W) & (Y[tb] >= 0) & (Y[tb] < H))[None]
            D = dx + dy + 1
            i, j = np.nonzero((D <= dmax[g - 1]) & (dx * dx + dy * dy < 250000) & inb)
            tails.append(offs[t] + i); heads.append(offs[tb] + j); gaps.append(np.full(len(i), g)); lens.append(D[i, j])
    a, b, g, L = (np.concatenate(v) for v in (tails, heads, gaps, lens))
    order = np.lexsort((b, g, a))
    a, b, g, L = a[order], b[order], g[order], L[order]
    cost = _arc_cost_int_vec(table[g - 1, L], 3, a, b)
    n = int(offs[-1])
    row_ptr = np.zeros(n + 1, np.int64)
    row_ptr[1:] = np.cumsum(np.bincount(a, minlength=n))
    cf = np.concatenate([conf[t, :cnt[t]] for t in range(F)]).astype(np.float64)
    obs = orc.observation_cost(orc.cap_conf(cf, P['MCF_CONF_CAPPING_METHOD']), P['MCF_MAX_CONF_COST'])
    k = np.arange(n)
    ee = float(P['MCF_ENTRY_EXIT_COST'])
    obs_i, en_i, ex_i = (_arc_cost_int_vec(obs, 2, k, 0), _arc_cost_int_vec(np.full(n, ee), 0, k, 0),
                         _arc_cost_int_vec(np.full(n, ee), 1, k, 0))
    dets = [(conf[t, :cnt[t]], X[t], Y[t]) for t in range(F)]
    return obs_i, en_i, ex_i, row_ptr, b.astype(np.int32), cost, offs, dets


def c3_network():
    """The flow network of the config-3 bench timelapse (252 frames, 19 340 detections captured from the GPU path into
    tests/data/c3_dets.npz): (obs, entry, exit i64 [n], row_ptr i64 [n+1], col i32, cost i64, offs, dets)."""
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = np.load(os.path.join(ROOT, 'tests', 'data', 'c3_dets.npz'))
    return open_grid_network(d['count'], d['x'], d['y'], d['conf'], 512, 512)


def moving_network(n_frames, size, n_alive, seed=0, **kw):
    """The flow network of a scene of moving growth cones (synth.synth_detections): association-only workloads."""
    from axtrack_amd import synth
    d = synth.synth_detections(n_frames, size, size, n_alive=n_alive, seed=seed, **kw)
    return open_grid_network(d['count'], d['x'], d['y'], d['conf'], size, size)


def check_flow_certificate(obs, entry, exit_, row_ptr, col, cost, nxt, track, total_cost, pots, min_flow, max_flow):
    """Independent proof that (nxt, track) is a MINIMUM-cost flow of the tracker's network: primal feasibility (node-disjoint
    paths along existing arcs, flow count within bounds), the stated total cost, and complementary slackness of the node
    potentials `pots` = (pot_u, pot_v, pot_t) over ALL arcs (include/axtrack_hip.h, axt_mcf_solve_duals): reduced cost >= 0 on
    every arc without flow, <= 0 on every arc with flow. O(arcs) numpy, no second solve. Returns a dict of counts; raises
    AssertionError with the first violated condition."""
    obs, entry, exit_, row_ptr, cost = (np.asarray(a, np.int64) for a in (obs, entry, exit_, row_ptr, cost))
    col, nxt, track = np.asarray(col, np.int64), np.asarray(nxt, np.int64), np.asarray(track, np.int64)
    pu, pv, pt = np.asarray(pots[0], np.int64), np.asarray(pots[1], np.int64), int(pots[2])
    n = len(obs)
    used = track >= 0
    # ---- primal: every used detection has at most one successor / predecessor, successors are arcs of the network
    has_next = nxt >= 0
    assert not np.any(has_next & ~used), 'an unused detection has a successor'
    assert np.all(used[nxt[has_next]]), 'a successor is an unused detection'
    assert np.all(track[nxt[has_next]] == track[has_next]), 'a link joins two different trajectories'
    indeg = np.bincount(nxt[has_next], minlength=n)
    assert indeg.max(initial=0) <= 1, 'a detection has two predecessors'
    start = used & (indeg == 0)
    end = used & ~has_next
    F = int(start.sum())
    assert F == int(end.sum()) and F == (int(track.max()) + 1 if used.any() else 0), 'trajectory count'
    assert min_flow <= F <= max_flow, f'flow count {F} outside [{min_flow}, {max_flow}]'
    tail = np.repeat(np.arange(n, dtype=np.int64), np.diff(row_ptr))
    on = nxt[tail] == col                                            # arcs that carry flow
    assert int(on.sum()) == int(has_next.sum()), 'a link is not an arc of the network (or the network has parallel arcs)'
    total = int(obs[used].sum() + entry[start].sum() + exit_[end].sum() + cost[on].sum())
    assert total == int(total_cost), f'total cost {total_cost} stated, {total} recomputed'
    # ---- dual: complementary slackness, arc class by arc class
    def cs(rc, flow, what):
        assert np.all(rc[~flow] >= 0), f'{what}: negative reduced cost on an arc without flow ({int((rc[~flow] < 0).sum())} arcs)'
        assert np.all(rc[flow] <= 0), f'{what}: positive reduced cost on an arc with flow ({int((rc[flow] > 0).sum())} arcs)'
    cs(entry - pu, start, 'entry arcs')
    cs(obs + pu - pv, used, 'observation arcs')
    cs(exit_ + pv - pt, end, 'exit arcs')
    cs(cost + pv[tail] - pu[col], on, 'transition arcs')
    if F < max_flow:
        assert pt >= 0, f'one more trajectory would pay: pi(T) = {pt} < 0 with the flow count below max_flow'
    if F > min_flow:
        assert pt <= 0, f'one trajectory fewer would pay: pi(T) = {pt} > 0 with the flow count above min_flow'
    return {'arcs': int(len(col)), 'arcs_with_flow': int(on.sum()), 'trajectories': F, 'used': int(used.sum())}
