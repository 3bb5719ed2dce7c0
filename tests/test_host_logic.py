"""CPU-side tests of the product's host logic and of the C-ABI library (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

from axtrack_amd import _lib, params
from axtrack_amd import hotpath as hp
from axtrack_amd.detections import transition_cost_table, _arc_cost_int_vec
from oracle import oracle as orc
from helpers import golden_dets, csr_arcs_from_oracle, node_costs_from_oracle, tracks_from_next

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = set()
    for fn in os.listdir(os.path.join(ROOT, 'include')):
        if fn.endswith('.h'):
            text = open(os.path.join(ROOT, 'include', fn)).read()
            text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
            names |= set(re.findall(r'\b(axt_[a-z0-9_]+)\s*\(', text))
    assert len(names) >= 14
    for n in sorted(names):
        assert hasattr(lib, n), f'libaxtrack_hip.so does not export {n}'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in axtrack_amd/_lib.py'
    assert lib.axt_abi_version() == 1


def test_product_has_no_oracle_import():
    """The oracle is test infrastructure: nothing under axtrack_amd/ may import or load it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'axtrack_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                for pat in (r'\bimport\s+oracle', r'\bfrom\s+oracle', r'liboracle', r'oracle[/\\]', r"['\"]oracle['\"]",
                            r'orc\.'):
                    assert not re.search(pat, text), f'{f} reaches for the oracle ({pat})'


def test_arc_cost_int_three_implementations_agree():
    lib = _lib.load()
    rng = np.random.default_rng(0)
    cost = rng.uniform(-4.6, 7.0, 200)
    a = rng.integers(0, 2 ** 20, 200)
    b = rng.integers(0, 2 ** 20, 200)
    for kind in range(4):
        vec = _arc_cost_int_vec(cost, kind, a, b)
        for i in range(200):
            ref = orc.arc_cost_int(cost[i], kind, a[i], b[i])
            assert ref == lib.axt_arc_cost_int(float(cost[i]), kind, int(a[i]), int(b[i])) == int(vec[i])


def test_transition_table_matches_reference_costs(golden):
    a = golden('assoc_parts')
    table, dmax = transition_cost_table(params.DEPLOYED)
    assert list(dmax) == [251, 86]
    for g in (1, 2):
        assert np.array_equal(table[g - 1, 1:], a[f'trans_cost_gap{g}'])
    assert np.isinf(table[:, 500]).all()


def _solve(dets, H, W, P=orc.DEFAULTS):
    row_ptr, col, length, gap, cost, offs = csr_arcs_from_oracle(dets, H, W, P)
    obs_i, en_i, ex_i, _ = node_costs_from_oracle(dets, P)
    res = hp.mcf_solve(obs_i, en_i, ex_i, row_ptr, col, cost, P['MCF_MIN_FLOW'], P['MCF_MAX_FLOW'])
    return res, offs


def test_mcf_solver_matches_oracle_on_golden_detections(golden):
    g = golden('detect_1024')
    dets = golden_dets(g)
    D = orc.all_path_matrices(dets, 1024, 1024)
    trajs, total = orc.mcf_solve(dets, D)
    res, offs = _solve(dets, 1024, 1024)
    assert res is not None
    nxt, track, n_tracks, tot = res
    assert tot == total and n_tracks == len(trajs)
    assert tracks_from_next(nxt, track, offs) == trajs


@pytest.mark.parametrize('seed', range(6))
def test_mcf_solver_random_tracking_graphs(seed):
    """Random small timelapses: product solver == Bellman-Ford oracle (tracks and cost), and the
    optimum cost equals networkx' network simplex on the same network with the flow value fixed."""
    import networkx as nx
    rng = np.random.default_rng(seed)
    F = int(rng.integers(3, 9))
    dets = []
    for _ in range(F):
        n = int(rng.integers(0, 9))
        conf = np.sort(rng.uniform(0.55, 1.3, n).astype(np.float32))[::-1]
        dets.append((conf, rng.integers(0, 400, n), rng.integers(0, 400, n)))
    P = dict(orc.DEFAULTS, MCF_MIN_FLOW=int(rng.integers(0, 3)), MCF_MAX_FLOW=int(rng.integers(3, 12)))
    if sum(len(d[0]) for d in dets) == 0:
        return
    D = orc.all_path_matrices(dets, 400, 400)
    trajs, total = orc.mcf_solve(dets, D, P)
    res, offs = _solve(dets, 400, 400, P)
    if trajs is None:
        assert res is None
        return
    nxt, track, n_tracks, tot = res
    assert tot == total
    assert tracks_from_next(nxt, track, offs) == trajs
    # independent optimum check
    tail, head, cost, _ = orc.build_flow_graph(dets, D, P)
    G = nx.DiGraph()
    Fv = len(trajs)
    G.add_node(0, demand=-Fv); G.add_node(1, demand=Fv)
    for t, h, c in zip(tail, head, cost):
        G.add_edge(int(t), int(h), capacity=1, weight=int(c))
    assert nx.min_cost_flow_cost(G) == total


@pytest.mark.parametrize('seed', range(12))
def test_mcf_solver_time_blocks_on_random_timelapses(seed, monkeypatch):
    """The time-blocked assignment solver on networks small enough for the oracle, with leaves forced down to a
    handful of detections so that blocks, separators (also empty ones, and ones the reach test rejects) and their
    joins all occur: trajectories and cost equal the oracle's at every thread count."""
    rng = np.random.default_rng(100 + seed)
    F = int(rng.integers(8, 30))
    dets = []
    for t in range(F):
        n = int(rng.integers(0, 7)) if seed % 3 else int(rng.integers(3, 10))
        conf = np.sort(rng.uniform(0.55, 1.3, n).astype(np.float32))[::-1]
        dets.append((conf, rng.integers(0, 300, n), rng.integers(0, 300, n)))
    if sum(len(d[0]) for d in dets) == 0:
        return
    P = dict(orc.DEFAULTS, MCF_MIN_FLOW=0, MCF_MAX_FLOW=1000)
    D = orc.all_path_matrices(dets, 300, 300)
    trajs, total = orc.mcf_solve(dets, D, P)
    monkeypatch.setenv('AXT_MCF_MIN_LEAF', str(int(rng.integers(1, 12))))
    for threads in ('1', '2', '4', '16'):
        monkeypatch.setenv('AXT_MCF_THREADS', threads)
        res, offs = _solve(dets, 300, 300, P)
        nxt, track, n_tracks, tot = res
        assert tot == total and tracks_from_next(nxt, track, offs) == trajs


def test_mcf_infeasible_and_argument_errors():
    dets = [(np.array([0.9], np.float32), np.array([10]), np.array([10]))]
    P = dict(orc.DEFAULTS, MCF_MIN_FLOW=5)
    res, _ = _solve(dets, 100, 100, P)
    assert res is None                                     # fewer than min_flow trajectories exist
    lib = _lib.load()
    bad_col = np.array([0], np.int32)                      # arc pointing backwards in time
    z = np.zeros(1, np.int64)
    rp = np.array([0, 1], np.int64)
    out = np.zeros(1, np.int32)
    n, tot = ctypes.c_int(), ctypes.c_int64()
    rc = lib.axt_mcf_solve(1, z.ctypes.data, z.ctypes.data, z.ctypes.data, rp.ctypes.data, bad_col.ctypes.data,
                           z.ctypes.data, 0, 1, out.ctypes.data, out.ctypes.data, ctypes.byref(n), ctypes.byref(tot))
    assert rc == -22 and b'forward in time' in lib.axt_last_error()


def test_flow_solver_fast_and_general_paths_agree_at_full_size(monkeypatch):
    """The detections of the C3 bench timelapse (252 frames, 19 340 detections, captured from the GPU path into
    tests/data/c3_dets.npz): the assignment-form solver and the successive-shortest-path solver must return the same
    optimum and the same trajectories on the full 553 k-arc network, so must the assignment solver at any number of
    threads, and all of them the oracle's own solver on the oracle's own network."""
    from helpers import c3_network
    obs_i, en_i, ex_i, row_ptr, b, cost, offs, dets = c3_network()
    assert len(b) == 553073
    monkeypatch.setenv('AXT_MCF_THREADS', '1')
    fast = hp.mcf_solve(obs_i, en_i, ex_i, row_ptr, b, cost, 5, 450)
    # blocks of frames solved on concurrent threads and joined through the rows between them: the same optimum
    for threads in ('2', '8', '16'):
        monkeypatch.setenv('AXT_MCF_THREADS', threads)
        par = hp.mcf_solve(obs_i, en_i, ex_i, row_ptr, b, cost, 5, 450)
        assert par[2] == fast[2] and par[3] == fast[3] and np.array_equal(par[0], fast[0]) and np.array_equal(par[1], fast[1])
    # the oracle's own network and solver (Bellman-Ford successive shortest paths, ~35 s) at this size
    trajs, total = orc.mcf_solve(dets, orc.all_path_matrices(dets, 512, 512), dict(orc.DEFAULTS))
    assert total == fast[3] and tracks_from_next(fast[0], fast[1], offs) == trajs
    monkeypatch.setenv('AXT_MCF_FORCE_SSP', '1')
    slow = hp.mcf_solve(obs_i, en_i, ex_i, row_ptr, b, cost, 5, 450)
    assert fast[2] == slow[2] == 63 and fast[3] == slow[3]
    assert np.array_equal(fast[0], slow[0]) and np.array_equal(fast[1], slow[1])
    # a flow bound that the unconstrained optimum violates exercises the fallback from the fast path
    monkeypatch.delenv('AXT_MCF_FORCE_SSP')
    capped = hp.mcf_solve(obs_i, en_i, ex_i, row_ptr, b, cost, 5, 40)
    assert capped[2] == 40 and capped[3] > fast[3]


def test_sharded_ided_blocks_assemble_to_the_global_table():
    """Frame-sharded runs leave every rank with its block of IDed_dets_all (its frames x the identities alive there);
    sharded.assemble_ided_dets_all must rebuild the single-process table, label quirk included (a frame without
    any IDed detection in the middle of rank 1's block, an identity that lives on both ranks, one that does not)."""
    import pandas as pd
    from axtrack_amd import sharded
    from axtrack_amd.detections import _axon_index
    F = 8
    # per frame: list of (id, conf, x, y)
    tables = [[(0, .9, 10, 11), (1, .8, 20, 21)], [(0, .7, 12, 13)], [(1, .6, 22, 23), (2, .95, 30, 31)], [(2, .9, 32, 33)],
              [(2, .85, 34, 35), (3, .75, 40, 41)], [], [(3, .7, 42, 43)], [(3, .65, 44, 45), (4, .99, 50, 51)]]
    for quirk in (True, False):
        ids, labels, info, ref = orc.ided_dets_all(tables, reproduce_label_quirk=quirk)
        blocks = []
        for a, b in ((0, 4), (4, 8)):
            alive = sorted({r[0] for f in range(a, b) for r in tables[f]})
            v = np.full((len(alive), 3 * (b - a)), np.nan)
            for f in range(a, b):
                for tid, c, x, y in tables[f]:
                    v[alive.index(tid), 3 * (f - a):3 * (f - a) + 3] = (x, y, c)
            cols = pd.MultiIndex.from_product([range(a, b), ['anchor_x', 'anchor_y', 'conf']])
            blocks.append(pd.DataFrame(v, index=_axon_index(np.array(alive)), columns=cols))
        whole = sharded.assemble_ided_dets_all(blocks[::-1], F, quirk)          # order of arrival must not matter
        assert list(whole.index) == [f'Axon_{i:0>3}' for i in ids]
        assert np.array_equal(np.nan_to_num(whole.to_numpy(), nan=-1), np.nan_to_num(ref, nan=-1))
        assert [c[0] for c in whole.columns] == list(labels.astype(int))


def _pure_cost_of(solution, obs, en, ex, row_ptr, col, cost):
    """Cost of a solution (next, track) in whole cost units, i.e. without the identity hash in the low 16 bits."""
    nxt, track = solution[0], solution[1]
    used = track >= 0
    has_pred = np.zeros(len(nxt), bool)
    has_pred[nxt[nxt >= 0]] = True
    tot = int((obs[used] >> 16).sum() + (en[used & ~has_pred] >> 16).sum() + (ex[used & (nxt < 0)] >> 16).sum())
    n_arcs = int(used.sum()) + int((used & ~has_pred).sum()) + int((used & (nxt < 0)).sum())
    for k in np.nonzero(nxt >= 0)[0]:
        lo, hi = row_ptr[k], row_ptr[k + 1]
        tot += int(cost[lo + int(np.nonzero(col[lo:hi] == nxt[k])[0][0])] >> 16)
        n_arcs += 1
    return tot, n_arcs


def test_identity_hash_does_not_change_the_optimum_of_the_unperturbed_problem(golden):
    """Every integer arc cost is round(cost * 1e6) * 2^16 + hash16(arc identity): the hash makes the optimum unique, but it
    is not a strict secondary key -- summed over the A flow-carrying arcs of a solution it can reach A * (1 - 2^-16)
    cost units, so the perturbed optimum is only guaranteed to be within A units (1e-6 each) of the optimum of the
    unperturbed network, which is what the reference hands to libmot. Measured here: on the reference-generated golden
    detections (against networkx' network simplex on the UNPERTURBED network) and on the config-3 network (against
    this solver on the unperturbed costs) the gap is zero -- the perturbed optimum is an optimum of the unperturbed
    problem -- and the bound holds on random graphs."""
    import networkx as nx
    from helpers import c3_network
    # golden detections (922, 3 frames) -- independent solver on the unperturbed network
    dets = golden_dets(golden('detect_1024'))
    P = dict(orc.DEFAULTS, MCF_MIN_FLOW=0)
    row_ptr, col, length, gap, cost, offs = csr_arcs_from_oracle(dets, 1024, 1024, P)
    obs_i, en_i, ex_i, _ = node_costs_from_oracle(dets, P)
    res = hp.mcf_solve(obs_i, en_i, ex_i, row_ptr, col, cost, 0, 450)
    pure, n_arcs = _pure_cost_of(res, obs_i, en_i, ex_i, row_ptr, col, cost)
    G = nx.DiGraph()
    G.add_node('S', demand=-res[2]); G.add_node('T', demand=res[2])
    for k in range(len(obs_i)):
        G.add_edge('S', ('u', k), capacity=1, weight=int(en_i[k] >> 16))
        G.add_edge(('u', k), ('v', k), capacity=1, weight=int(obs_i[k] >> 16))
        G.add_edge(('v', k), 'T', capacity=1, weight=int(ex_i[k] >> 16))
        for e in range(row_ptr[k], row_ptr[k + 1]):
            G.add_edge(('v', k), ('u', int(col[e])), capacity=1, weight=int(cost[e] >> 16))
    best = nx.min_cost_flow_cost(G)
    assert 0 <= pure - best <= n_arcs
    assert pure == best, f'the hash moved the golden optimum by {pure - best} units'
    # config 3 (19 340 detections, 553 k arcs): same flow count, same cost, without the hash
    obs_i, en_i, ex_i, row_ptr, col, cost, offs, _ = c3_network()
    res = hp.mcf_solve(obs_i, en_i, ex_i, row_ptr, col, cost, 5, 450)
    unpert = hp.mcf_solve((obs_i >> 16) << 16, (en_i >> 16) << 16, (ex_i >> 16) << 16, row_ptr, col, (cost >> 16) << 16, 5, 450)
    pure, n_arcs = _pure_cost_of(res, obs_i, en_i, ex_i, row_ptr, col, cost)
    assert res[2] == unpert[2] == 63
    assert pure == unpert[3] >> 16, f'the hash moved the config-3 optimum by {pure - (unpert[3] >> 16)} units (bound {n_arcs})'


@pytest.mark.parametrize('seed', range(6))
def test_flow_solver_one_and_two_phase_schedules_agree(seed, monkeypatch):
    """The assignment solver's two-phase schedule (exits at a fifth of their price first, then the track ends re-inserted at
    the full price) and the one-phase schedule return the oracle's trajectories and cost -- dense random timelapses where
    tracks dissolve and merge between the phases, with flow bounds and time blocks."""
    rng = np.random.default_rng(900 + seed)
    F = int(rng.integers(6, 24))
    dets = []
    for t in range(F):
        n = int(rng.integers(2, 11))
        conf = np.sort(rng.uniform(0.55, 1.3, n).astype(np.float32))[::-1]
        dets.append((conf, rng.integers(0, 150, n), rng.integers(0, 150, n)))
    P = dict(orc.DEFAULTS, MCF_MIN_FLOW=int(rng.integers(0, 3)), MCF_MAX_FLOW=int(rng.integers(3, 40)),
             MCF_ENTRY_EXIT_COST=float(rng.choice([0.3, 1.0, 2.0])))
    trajs, total = orc.mcf_solve(dets, orc.all_path_matrices(dets, 400, 400), P)
    monkeypatch.setenv('AXT_MCF_MIN_LEAF', '4')
    for env in ('AXT_MCF_ONE_PHASE', 'AXT_MCF_TWO_PHASE'):
        monkeypatch.delenv('AXT_MCF_ONE_PHASE', raising=False)
        monkeypatch.delenv('AXT_MCF_TWO_PHASE', raising=False)
        monkeypatch.setenv(env, '1')
        for threads in ('1', '8'):
            monkeypatch.setenv('AXT_MCF_THREADS', threads)
            res, offs = _solve(dets, 400, 400, P)
            if trajs is None:
                assert res is None
                continue
            assert res[3] == total and tracks_from_next(res[0], res[1], offs) == trajs


def test_flow_solver_on_a_scene_of_moving_cones_with_concurrent_track_end_searches(monkeypatch):
    """An association-only workload (synth.synth_detections: cones that move, are born, die, are missed; clutter) at
    config 3's size: the second phase's track-end searches run side by side on shared rows and columns (claims,
    wait-die) and must reach the optimum of the serial schedule, of the one-phase schedule and of the successive-
    shortest-path solver -- same trajectories, same cost."""
    from helpers import moving_network
    net = moving_network(120, 512, 90, seed=3)[:6]
    monkeypatch.setenv('AXT_MCF_THREADS', '8')
    monkeypatch.setenv('AXT_MCF_ENDS_THREADS', '1')
    ref = hp.mcf_solve(*net, 5, 100000)
    assert ref[2] > 20
    for env in ({'AXT_MCF_ENDS_THREADS': '8'}, {'AXT_MCF_ENDS_THREADS': '3', 'AXT_MCF_MIN_LEAF': '64'}, {'AXT_MCF_ONE_PHASE': '1'},
                {'AXT_MCF_FORCE_SSP': '1'}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for _ in range(3 if 'AXT_MCF_ENDS_THREADS' in env else 1):       # the interleaving differs from run to run, the result must not
            got = hp.mcf_solve(*net, 5, 100000)
            assert got[2] == ref[2] and got[3] == ref[3] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        for k in env:
            monkeypatch.delenv(k)
        monkeypatch.setenv('AXT_MCF_ENDS_THREADS', '1')


@pytest.mark.parametrize('name', ['detect_1024', 'detect_ragged'])
def test_unstitched_tile_tables_equal_the_references(golden, name):
    """get_frame_dets(unstitched=True) / _pandas_tiled_dets (AxonDetections.py:178-248,322-331) decoded on the host from
    the YOLO grids: on the reference's golden grids the tables equal the reference's own (values and order)."""
    import torch
    from axtrack_amd.detections import AxonDetections
    g = golden(name)
    ad = object.__new__(AxonDetections)
    ad.Sx = ad.Sy = 12
    ad.tilesize = 512
    ad.conf_thr = 0.7
    ad.all_conf_thrs = np.sort(np.append(np.arange(0.55, 1, .04), 0.7)).round(2)
    ad._yolo = torch.from_numpy(g['yolo'])
    ad._tiled_tables = None
    ad.d_count = torch.zeros(g['yolo'].shape[0], dtype=torch.int32)
    tiled = g['tiled']
    n_checked = 0
    for t in range(g['yolo'].shape[0]):
        tabs = ad.get_frame_dets('all', t, unstitched=True)
        assert len(tabs) == g['yolo'].shape[1]
        for k, d in enumerate(tabs):
            ref = tiled[(tiled[:, 0] == t) & (tiled[:, 1] == k)]
            assert len(d) == len(ref)
            assert np.array_equal(d.conf.to_numpy(dtype=np.float64), ref[:, 2])
            assert np.array_equal(d.anchor_x.to_numpy(dtype=np.int64), ref[:, 3].astype(np.int64))
            assert np.array_equal(d.anchor_y.to_numpy(dtype=np.int64), ref[:, 4].astype(np.int64))
            assert str(d.conf.dtype) == 'Float32' and str(d.anchor_x.dtype) == 'Int64'
            n_checked += len(d)
    assert n_checked > 100


@pytest.mark.parametrize('world', [2, 4, 8])
def test_flow_solve_shared_between_ranks_in_one_process(world, monkeypatch):
    """axt_mcf_shard_begin / export / finish without a process group: `world` shard objects stand for the ranks; each solves
    its run of time blocks, the states are handed round, every one finishes with the single-process optimum -- on the
    config-3 network, a scene of moving cones, a network too small to share (every rank then solves it whole) and an
    infeasible one."""
    from helpers import c3_network, moving_network
    monkeypatch.setenv('AXT_MCF_THREADS', '2')
    nets = [(moving_network(120, 512, 90, seed=3)[:6], 5, 100000, None), (moving_network(6, 256, 8, seed=1)[:6], 0, 1000, None),
            (moving_network(6, 256, 8, seed=1)[:6], 500, 1000, 'infeasible')]
    if world == 4:
        nets.append((c3_network()[:6], 5, 450, None))
    for net, lo, hi, what in nets:
        ref = hp.mcf_solve(*net, lo, hi)
        shards = [hp.McfShard(*net, r, world) for r in range(world)]
        states = [s.state for s in shards]
        for s in shards:
            got = s.finish(states, lo, hi)
            if what == 'infeasible':
                assert ref is None and got is None
                continue
            assert got[2] == ref[2] and got[3] == ref[3] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    # a missing or mis-sized state is an error, not a wrong answer
    net = moving_network(120, 512, 90, seed=3)[:6]
    shards = [hp.McfShard(*net, r, 2) for r in range(2)]
    if len(shards[1].state):
        with pytest.raises(_lib.AxtError):
            shards[0].finish([shards[0].state, shards[1].state[:-16]], 5, 100000)
    with pytest.raises(_lib.AxtError):
        hp.McfShard(*net, 0, 3)


def test_moving_cone_scene_is_deterministic_and_obeys_the_detectors_rules():
    """synth.synth_detections (association-only workloads): the same arguments give the same bytes; every frame is in
    descending confidence, at or above the 0.55 floor, with no two detections closer than the NMS distance, anchors inside
    the frame; cones live for several frames and move by at most max_step + 2 jitter per frame and axis."""
    from axtrack_amd import synth
    a = synth.synth_detections(40, 512, 512, n_alive=60, seed=9)
    b = synth.synth_detections(40, 512, 512, n_alive=60, seed=9)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = synth.synth_detections(40, 512, 512, n_alive=60, seed=10)
    assert not np.array_equal(a['x'], c['x'])
    last = {}
    moved = []
    for t in range(40):
        n = int(a['count'][t])
        assert 20 < n <= a['conf'].shape[1]
        conf, x, y, tr = a['conf'][t, :n], a['x'][t, :n].astype(np.int64), a['y'][t, :n].astype(np.int64), a['truth'][t, :n]
        assert np.all(np.diff(conf.astype(np.float64)) <= 0) and conf.min() >= np.float32(0.55)
        assert x.min() >= 0 and x.max() < 512 and y.min() >= 0 and y.max() < 512
        d2 = (x[:, None] - x[None]) ** 2 + (y[:, None] - y[None]) ** 2
        np.fill_diagonal(d2, 10 ** 9)
        assert d2.min() >= 529
        for i, cone in enumerate(tr):
            if cone >= 0:
                if cone in last and last[cone][0] == t - 1:
                    moved.append(max(abs(int(x[i]) - last[cone][1]), abs(int(y[i]) - last[cone][2])))
                last[int(cone)] = (t, int(x[i]), int(y[i]))
    assert len(moved) > 500 and max(moved) <= 10 + 2 * 2 + 1


def test_bench_reads_the_committed_counters_of_the_newest_round():
    """bench.committed_counters: mfma_busy and hbm_gbps of the dominant kernel come from the newest committed profiles/*_pmc.csv
    / *_kernels.csv that hold it (file names sort by round), also from summaries whose kernel names hold unquoted commas."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    table = [dict(name=n, launches=l) for n, l in (('conv0 5>20 s2', 2), ('conv1 20>40 s2', 2), ('conv2 40>80 +pool', 2), ('conv4 80>80', 2),
                                                   ('conv5 80>80 +pool', 2), ('conv7 80>80', 1), ('conv8 80>80 +pool', 1), ('conv10 80>160', 1))]
    wino = ('conv2', 'conv4', 'conv5', 'conv7', 'conv8', 'conv10')
    got = bench.committed_counters(table, 'conv2 40>80 +pool', True, wino, 3.85)
    assert got['mfma_busy']['source'] == 'profiles/r04h_pmc.csv' and 0.6 < got['mfma_busy']['kernel'] < 0.8
    assert got['hbm_gbps']['source'] == 'profiles/r04h_kernels.csv' and got['traffic'] > 4e8
    # the two stride-2 blocks as one kernel: the bench's row 'conv0+1 ... fused' <-> the profile's conv_s2_fused
    fused_table = [dict(name='conv0+1 5>20>40 s2 fused', launches=2), dict(name='conv1 20>40 s2', launches=0)] + table[2:]
    f = bench.committed_counters(fused_table, 'conv0+1 5>20>40 s2 fused', True, wino, 3.6)
    assert f['mfma_busy']['source'] == 'profiles/r04h_pmc.csv' and 0.6 < f['mfma_busy']['kernel'] < 0.8
    assert 3e8 < f['traffic'] < 6e8 and f['hbm_gbps']['whole_cnn'] > 0
    assert bench.committed_counters(table, 'conv0 5>20 s2', True, wino, 3.85)['hbm_gbps']['source'] == 'profiles/r03z_kernels.csv'
    direct = bench.committed_counters(table, 'conv2 40>80 +pool', False, wino, 5.6)       # the direct kernels of the variant pass
    assert direct['mfma_busy']['kernel'] > got['mfma_busy']['kernel']
    old = bench._read_profile_csv(os.path.join(root, 'profiles', 'r02y_pmc.csv'))         # unquoted commas in the kernel column
    assert any(r['kernel'] == 'conv3x3_wino<40->80,s1,pool>' and float(r['GRBM_GUI_ACTIVE']) > 0 for r in old)


def test_flow_certificate_proves_the_optimum_and_rejects_anything_else(monkeypatch):
    """axt_mcf_solve_duals / axt_mcf_shard_finish_duals: the node potentials returned with the trajectories satisfy
    complementary slackness over ALL arcs (helpers.check_flow_certificate: O(arcs), no second solve) -- in every regime of the
    solver (assignment form with the flow count strictly inside its bounds; successive shortest paths with max_flow binding,
    with min_flow binding, with no further path at all), for the two-phase and the one-phase schedule, threaded, and for the
    solve shared between ranks. The checker has teeth: a worse flow, a flow with a detection on two tracks, wrong
    potentials and a wrong total each fail it."""
    from helpers import c3_network, moving_network, check_flow_certificate
    monkeypatch.setenv('AXT_MCF_THREADS', '4')
    c3 = c3_network()[:6]
    small = moving_network(30, 256, 12, seed=2)[:6]
    cases = [(c3, 5, 450), (c3, 5, 40), (c3, 80, 450), (small, 0, 1000), (small, 0, 3), (small, 40, 1000),
             (moving_network(60, 512, 60, seed=5)[:6], 5, 100000)]
    for net, lo, hi in cases:
        res = hp.mcf_solve(*net, lo, hi, duals=True)
        assert res is not None
        plain = hp.mcf_solve(*net, lo, hi)
        assert np.array_equal(res[0], plain[0]) and np.array_equal(res[1], plain[1]) and res[2:4] == plain[2:4]
        out = check_flow_certificate(*net, res[0], res[1], res[3], res[4], lo, hi)
        assert out['trajectories'] == res[2]
    # every detection used up: no further path exists (the branch that shifts the reached part of the network)
    tiny = moving_network(4, 128, 3, seed=1)[:6]
    full = hp.mcf_solve(*tiny, len(tiny[0]), 10 ** 6, duals=True)
    if full is not None:
        check_flow_certificate(*tiny, full[0], full[1], full[3], full[4], len(tiny[0]), 10 ** 6)
    # schedules and the forced successive-shortest-path solver
    for env in ({'AXT_MCF_ONE_PHASE': '1'}, {'AXT_MCF_TWO_PHASE': '1'}, {'AXT_MCF_FORCE_SSP': '1'}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = small if 'AXT_MCF_FORCE_SSP' not in env else moving_network(20, 256, 10, seed=4)[:6]
        res = hp.mcf_solve(*net, 0, 1000, duals=True)
        check_flow_certificate(*net, res[0], res[1], res[3], res[4], 0, 1000)
        for k in env:
            monkeypatch.delenv(k)
    # shared between ranks
    net = moving_network(120, 512, 90, seed=3)[:6]
    shards = [hp.McfShard(*net, r, 2) for r in range(2)]
    states = [s.state for s in shards]
    for s in shards:
        res = s.finish(states, 5, 100000, duals=True)
        check_flow_certificate(*net, res[0], res[1], res[3], res[4], 5, 100000)
    # ---- the checker rejects what is not the optimum
    net, lo, hi = c3, 5, 450
    nxt, track, n_tracks, total, pots = hp.mcf_solve(*net, lo, hi, duals=True)
    obs, en, ex, row_ptr, col, cost = net
    # (1) a feasible but worse flow: one trajectory removed, total recomputed honestly
    worse_track = np.where(track == n_tracks - 1, -1, track)
    worse_next = np.where(track == n_tracks - 1, -1, nxt)
    gone = track == n_tracks - 1
    tail = np.repeat(np.arange(len(obs)), np.diff(row_ptr))
    on = (nxt[tail] == col) & gone[tail]
    first = gone & (np.bincount(nxt[nxt >= 0], minlength=len(obs)) == 0)
    worse_total = total - int(obs[gone].sum() + cost[on].sum() + en[first].sum() + ex[gone & (nxt < 0)].sum())
    with pytest.raises(AssertionError, match='reduced cost'):
        check_flow_certificate(*net, worse_next, worse_track, worse_total, pots, lo, hi)
    # (2) not a flow: a detection gets a second predecessor
    k = int(np.flatnonzero(nxt >= 0)[0])
    other = int(np.flatnonzero((nxt >= 0) & (np.arange(len(nxt)) != k) & (track != track[k]))[0])
    bad_next = nxt.copy(); bad_next[other] = nxt[k]
    with pytest.raises(AssertionError):
        check_flow_certificate(*net, bad_next, track, total, pots, lo, hi)
    # (3) potentials that are not the optimum's, (4) a wrong total
    bad = (pots[0].copy(), pots[1].copy(), pots[2]); bad[1][k] += 1 << 30
    with pytest.raises(AssertionError, match='reduced cost'):
        check_flow_certificate(*net, nxt, track, total, bad, lo, hi)
    with pytest.raises(AssertionError, match='total cost'):
        check_flow_certificate(*net, nxt, track, total + 1, pots, lo, hi)


def test_shared_flow_solve_does_not_depend_on_a_ranks_thread_count_and_rejects_another_tree(monkeypatch):
    """The ranks of a shared solve must cut the timelapse into the same time blocks. The leaf count comes from rank-independent
    inputs only (a rank's thread budget -- its CPU set, AXT_MCF_THREADS -- no longer enters), and a state carries a hash of
    the cuts: a rank that built another tree (here: another AXT_MCF_MIN_LEAF) is refused instead of being imported."""
    from helpers import moving_network
    net = moving_network(120, 512, 90, seed=3)[:6]
    ref = hp.mcf_solve(*net, 5, 100000)
    monkeypatch.setenv('AXT_MCF_MIN_LEAF', '256')
    monkeypatch.setenv('AXT_MCF_THREADS', '1')
    s0 = hp.McfShard(*net, 0, 2)
    monkeypatch.setenv('AXT_MCF_THREADS', '7')
    s1 = hp.McfShard(*net, 1, 2)
    assert len(s0.state) > 16 and len(s1.state) > 16
    for s in (s0, s1):
        got = s.finish([s0.state, s1.state], 5, 100000)
        assert got[2] == ref[2] and got[3] == ref[3] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    monkeypatch.setenv('AXT_MCF_MIN_LEAF', '700')
    other = hp.McfShard(*net, 1, 2)
    s0b = hp.McfShard(*net, 0, 2)
    monkeypatch.setenv('AXT_MCF_MIN_LEAF', '256')
    s0c = hp.McfShard(*net, 0, 2)
    if len(other.state) == len(s1.state):                 # same size by chance: only the hash can tell
        with pytest.raises(_lib.AxtError, match='another time-block tree'):
            s0c.finish([s0c.state, other.state], 5, 100000)
    else:
        with pytest.raises(_lib.AxtError):
            s0c.finish([s0c.state, other.state], 5, 100000)
    del s0b


@pytest.mark.parametrize('threads', [2, 5])
def test_a_large_search_finished_by_a_team_of_threads_changes_nothing(threads, monkeypatch):
    """Search::finish_in_parallel (a search that has grown past a threshold goes on as a bucketed label-correcting search on
    the idle threads): forced onto small networks with a threshold of a few dozen rows, it ends with the trajectories, the
    cost and a valid optimality certificate of the serial solver -- static config-3 network (two phases, large second-phase
    searches), a scene of moving cones, one-phase schedule, and the solve shared between two ranks."""
    from helpers import c3_network, moving_network, check_flow_certificate
    nets = [(c3_network()[:6], 5, 450), (moving_network(100, 512, 90, seed=11)[:6], 5, 100000)]
    monkeypatch.setenv('AXT_MCF_THREADS', str(threads))
    for net, lo, hi in nets:
        monkeypatch.setenv('AXT_MCF_NO_PAR_SEARCH', '1')
        ref = hp.mcf_solve(*net, lo, hi)
        monkeypatch.delenv('AXT_MCF_NO_PAR_SEARCH')
        monkeypatch.setenv('AXT_MCF_PAR_MIN_N', '0')
        for switch, buckets, extra in ((48, 7, {}), (300, 48, {'AXT_MCF_ONE_PHASE': '1'})):
            monkeypatch.setenv('AXT_MCF_PAR_SWITCH', str(switch))
            monkeypatch.setenv('AXT_MCF_PAR_BUCKETS', str(buckets))
            for k, v in extra.items():
                monkeypatch.setenv(k, v)
            got = hp.mcf_solve(*net, lo, hi, duals=True)
            for k in extra:
                monkeypatch.delenv(k)
            assert got[2] == ref[2] and got[3] == ref[3] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
            check_flow_certificate(*net, got[0], got[1], got[3], got[4], lo, hi)
        shards = [hp.McfShard(*net, r, 2) for r in range(2)]
        res = shards[1].finish([s.state for s in shards], lo, hi)
        assert res[3] == ref[3] and np.array_equal(res[1], ref[1])
        monkeypatch.delenv('AXT_MCF_PAR_MIN_N')

def test_flow_solver_on_forty_random_scenes_equals_its_other_regime_and_carries_a_certificate(monkeypatch):
    """Forty scenes of moving cones of random length and density, the time-block tree forced onto them with a random leaf size:
    the assignment-form solver (searches on a monotone radix heap: a search never pushes a key below the last one it popped, which
    is what the heap's buckets rely on) ends with the trajectories and the cost of the successive-shortest-path regime, and its
    duals certify the optimum."""
    from helpers import moving_network, check_flow_certificate
    for seed in range(40):
        rng = np.random.default_rng(seed)
        net = moving_network(int(rng.integers(20, 90)), 512, int(rng.integers(10, 80)), seed=seed)[:6]
        monkeypatch.setenv('AXT_MCF_MIN_LEAF', str(int(rng.integers(16, 400))))
        got = hp.mcf_solve(*net, 5, 100000, duals=True)
        monkeypatch.setenv('AXT_MCF_FORCE_SSP', '1')
        ref = hp.mcf_solve(*net, 5, 100000)
        monkeypatch.delenv('AXT_MCF_FORCE_SSP')
        assert got[2] == ref[2] and got[3] == ref[3] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), seed
        check_flow_certificate(*net, got[0], got[1], got[3], got[4], 5, 100000)


def test_launch_attributes_are_remembered_per_device_not_per_process(tmp_path):
    """hipFuncSetAttribute acts on the CURRENT device: the launchers remember per device index what they have set
    (AxtOncePerDevice, csrc/axt_common.h) instead of in one process-wide flag, so a second device in one process gets its own
    attributes (a >64 KB LDS kernel launched on cuda:1 without them fails). The table itself, on the host: a device's bit is
    set by mark() only, other devices stay pending, indices beyond 63 are never cached; without a visible device every call
    is pending (nothing is remembered for a device that could not be asked)."""
    import subprocess
    src = tmp_path / 'once.cpp'
    src.write_text(r'''
#include "axt_common.h"
void axt_set_error(const char *, ...) {}
int main()
{
    AxtOncePerDevice once;
    int n = 0;
    const bool have = hipGetDeviceCount(&n) == hipSuccess && n > 0;
    if (!once.pending()) return 1;                       // nothing set yet
    if (!have) { once.mark(); if (!once.pending()) return 2; }      // no device: never remembered
    once.dev = 3; once.mark();
    if (once.done.load() != (have ? (1ull << 3) | 0ull : 1ull << 3) && once.done.load() != ((1ull << 3) | 1ull)) return 3;
    once.dev = 5;
    if ((once.done.load() >> 5) & 1) return 4;           // another device is still pending
    once.mark();
    if (!((once.done.load() >> 5) & 1) || !((once.done.load() >> 3) & 1)) return 5;
    once.dev = 64; const unsigned long long before = once.done.load(); once.mark();
    if (once.done.load() != before) return 6;            // beyond the table: set up on every call
    return 0;
}
''')
    exe = tmp_path / 'once'
    csrc = os.path.join(ROOT, 'axtrack_amd', 'csrc')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-x', 'c++', '-std=c++17', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', f'-I{csrc}', str(src), '-o', str(exe),
                           '-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,/opt/rocm/lib'])
    assert subprocess.run([str(exe)]).returncode == 0
