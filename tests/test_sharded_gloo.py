"""The multi-GPU exchange step on CPU: two processes (gloo) each hold the detections of their
frame block, one all-gather gives both the whole timelapse, and the replicated integer-cost solve
returns identical trajectories on both ranks (and the same as a single process)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from axtrack_amd import sharded, hotpath as hp
    from oracle import oracle as orc
    from helpers import golden_dets, csr_arcs_from_oracle, node_costs_from_oracle
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'detect_ragged.npz'))
    dets = golden_dets(g)                                   # 2 frames -> one per rank
    start, per = sharded.frame_block(len(dets), rank, world)
    cap = 640
    mine = dets[start:start + per]
    conf = torch.zeros((per, cap)); x = torch.zeros((per, cap), dtype=torch.int32); y = torch.zeros((per, cap), dtype=torch.int32)
    cnt = torch.zeros((per,), dtype=torch.int32)
    for f, (c, xx, yy) in enumerate(mine):
        n = len(c)
        conf[f, :n] = torch.from_numpy(c.copy()); x[f, :n] = torch.from_numpy(xx.astype(np.int32))
        y[f, :n] = torch.from_numpy(yy.astype(np.int32)); cnt[f] = n
    gc, gx, gy, gn = sharded.all_gather_detections(conf, x, y, cnt)
    all_dets = [(gc[f, :gn[f]].numpy(), gx[f, :gn[f]].numpy().astype(np.int64), gy[f, :gn[f]].numpy().astype(np.int64))
                for f in range(world * per)]
    ok_gather = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
                    for a, b in zip(all_dets, dets))
    P = dict(orc.DEFAULTS, MCF_MIN_FLOW=1)
    row_ptr, col, length, gap, cost, offs = csr_arcs_from_oracle(all_dets, int(g['H']), int(g['W']), P)
    # sharded arc build: every rank holds the rows of its own frames, all_gather_arcs must rebuild the global list
    n_det = int(offs[-1])
    lo, hi = int(offs[start]), int(offs[start + per])
    a0, a1 = int(row_ptr[lo]), int(row_ptr[hi])
    local_ptr = np.zeros(n_det + 1, np.int64)
    local_ptr[lo:hi + 1] = row_ptr[lo:hi + 1] - a0
    local_ptr[hi + 1:] = a1 - a0
    g_ptr, g_col, g_len, g_gap, g_cost = sharded.all_gather_arcs(
        torch.from_numpy(local_ptr), torch.from_numpy(np.ascontiguousarray(col[a0:a1], np.int32)),
        torch.from_numpy(np.ascontiguousarray(length[a0:a1], np.int16)), torch.from_numpy(np.ascontiguousarray(gap[a0:a1], np.uint8)),
        torch.from_numpy(np.ascontiguousarray(cost[a0:a1], np.int64)), n_det)
    ok_gather = ok_gather and (np.array_equal(g_ptr.numpy(), row_ptr[:n_det + 1]) and np.array_equal(g_col.numpy(), col)
                               and np.array_equal(g_len.numpy(), length) and np.array_equal(g_gap.numpy(), gap)
                               and np.array_equal(g_cost.numpy(), cost) and len(col) > 0)
    obs_i, en_i, ex_i, _ = node_costs_from_oracle(all_dets, P)
    nxt, track, n_tracks, total = hp.mcf_solve(obs_i, en_i, ex_i, row_ptr, col, cost, 1, 450)
    q.put((rank, ok_gather, n_tracks, total, track.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_replicated_solve():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], 'gathered detections differ from the global list'
    assert res[0][2:] == res[1][2:], 'ranks disagree on the replicated solve'
    assert res[0][2] >= 1


def _flow_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), AXT_MCF_THREADS='2', AXT_MCF_MIN_LEAF='256')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from axtrack_amd import sharded, hotpath as hp
    from helpers import moving_network
    out = []
    for F, alive, bounds in ((96, 60, (5, 100000)), (40, 30, (0, 12))):       # the second: the flow bound binds (fallback path)
        net = moving_network(F, 512, alive, seed=5)[:6]
        sharded.COLLECTIVE_MS = {}
        res = sharded.solve_flow(*net, *bounds)
        names = sorted(sharded.COLLECTIVE_MS)
        sharded.COLLECTIVE_MS = None
        ref = hp.mcf_solve(*net, *bounds)                                   # the single-process solve of the same network
        same = res[2] == ref[2] and res[3] == ref[3] and np.array_equal(res[0], ref[0]) and np.array_equal(res[1], ref[1])
        out.append((same, res[2], res[3], res[1].tobytes(), names))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_flow_solve():
    """sharded.solve_flow (axt_mcf_shard_*): each of two ranks solves its run of time blocks, one all-gather exchanges the
    states, both join them: trajectories and cost of the single-process solve on both ranks, also when the flow bound binds
    and the solver falls back to successive shortest paths. Exactly two collectives (state sizes, states)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0], res[1]):
        assert a[0] and b[0], 'the sharded solve differs from the single-process solve'
        assert a[1:4] == b[1:4], 'ranks disagree'
        assert a[4] == ['flow_state_sizes_allreduce', 'flow_states_allgather']
    assert res[0][0][1] > 10 and res[0][1][1] == 12


def test_frame_block_partition():
    from axtrack_amd import sharded
    assert [sharded.frame_block(1008, r, 8) for r in (0, 7)] == [(0, 126), (882, 126)]
    import pytest
    with pytest.raises(ValueError):
        sharded.frame_block(10, 0, 4)
