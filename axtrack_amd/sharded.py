"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed).

Detection is embarrassingly parallel over frames: rank r owns a contiguous block of detection
frames and reads a +-2-frame input halo; no communication. Association is global: ONE all-gather of the
per-frame detection lists (a few MB: latency-bound; RCCL over xGMI on GPUs, gloo on CPU in the tests) gives every
rank all detections. The per-frame work of the association stays sharded -- the Hungarian variant's frame pairs
(one MAX all-reduce joins the links), the flow tracker's arc rows with their path searches (all_gather_arcs) -- and
the flow solve itself is shared too (solve_flow: every rank solves its run of time blocks, one all-gather of the
runs' states, then every rank joins them -- a unique optimum: no broadcast needed).
"""
import time

import torch
import torch.distributed as dist

COLLECTIVE_MS = None        # bench.py: set to a dict to collect the wall time of every collective (synchronised; off by default)


def _collective(name, fn):
    """Run one collective; when COLLECTIVE_MS is a dict, bracket it with device synchronisations and record its wall time."""
    if COLLECTIVE_MS is None:
        return fn()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    COLLECTIVE_MS[name] = COLLECTIVE_MS.get(name, 0.0) + (time.perf_counter() - t) * 1e3
    return r


def frame_block(n_frames_total, rank, world):
    """Contiguous, equal blocks (the caller pads n_frames_total to a multiple of world)."""
    if n_frames_total % world:
        raise ValueError(f'{n_frames_total} detection frames do not split evenly over {world} ranks')
    per = n_frames_total // world
    return rank * per, per


def all_gather_detections(conf, x, y, count, group=None, check_shapes=True):
    """conf f32 [F,cap], x/y i32 [F,cap], count i32 [F] per rank -> the same arrays for all
    world*F frames, rank-major. One collective on a packed i32 buffer [F, 3*cap+1]."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return conf, x, y, count
    world = dist.get_world_size(group)
    F, cap = conf.shape
    # every rank must bring the same [F, cap]: cap follows from the kept-tile list, which is a property of the whole
    # timelapse (Timelapse.sync_tile_occupancy), F from the equal frame blocks. A mismatch would hang or corrupt the
    # gather, so it is checked first (one tiny collective, once per timelapse: AxonDetections.gather_detections).
    shape = torch.tensor([F, cap, -F, -cap], dtype=torch.int64, device=conf.device)
    if check_shapes:
        _collective('shape_check_allreduce', lambda: dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=group))
    if check_shapes and (shape[0].item() != -shape[2].item() or shape[1].item() != -shape[3].item()):
        raise ValueError(f'frame-sharded ranks disagree on the detection array shape (this rank: [{F}, {cap}]; '
                         f'max [{shape[0].item()}, {shape[1].item()}]): call Timelapse.sync_tile_occupancy() on every '
                         f'rank before detect_dataset() and give every rank the same number of frames')
    packed = torch.cat([conf.contiguous().view(torch.int32), x, y, count.view(F, 1)], dim=1).contiguous()
    out = torch.empty((world * F, 3 * cap + 1), dtype=torch.int32, device=packed.device)
    _collective('detections_allgather', lambda: dist.all_gather_into_tensor(out, packed, group=group))
    g_conf = out[:, :cap].contiguous().view(torch.float32)
    g_x = out[:, cap:2 * cap].contiguous()
    g_y = out[:, 2 * cap:3 * cap].contiguous()
    g_count = out[:, 3 * cap].contiguous()
    return g_conf, g_x, g_y, g_count


def all_gather_arcs(row_ptr, col, length, gap, cost, n_det, group=None):
    """Every rank has built the arc rows of its own frames (hotpath.build_arcs(src_count=...): row_ptr covers all n_det
    detections, rows of other ranks' frames are empty). Returns the arc list of the whole timelapse on every rank --
    the ranks' lists in rank order, which is row order because the frame blocks are contiguous. Two collectives: a SUM
    all-reduce of the per-row counts (with the list lengths in its tail) and one all-gather of the arcs, packed as two
    i64 words each (cost | col, length, gap) and padded to the longest list."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = col.device
    n = int(col.numel())
    counts = torch.zeros((n_det + world,), dtype=torch.int64, device=dev)
    counts[:n_det] = row_ptr[1:n_det + 1] - row_ptr[:n_det]
    counts[n_det + rank] = n
    _collective('arc_counts_allreduce', lambda: dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group))
    sizes = counts[n_det:].tolist()
    longest = max(max(sizes), 1)
    packed = torch.zeros((longest, 2), dtype=torch.int64, device=dev)
    packed[:n, 0] = cost
    packed[:n, 1] = col.to(torch.int64) | (length.to(torch.int64) << 32) | (gap.to(torch.int64) << 48)
    out = torch.empty((world * longest, 2), dtype=torch.int64, device=dev)
    _collective('arcs_allgather', lambda: dist.all_gather_into_tensor(out, packed, group=group))
    parts = [out[r * longest:r * longest + sizes[r]] for r in range(world)]
    allp = torch.cat(parts, dim=0)
    g_row_ptr = torch.zeros((n_det + 1,), dtype=torch.int64, device=dev)
    g_row_ptr[1:] = torch.cumsum(counts[:n_det], 0)
    w = allp[:, 1]
    return (g_row_ptr, (w & 0xffffffff).to(torch.int32), ((w >> 32) & 0xffff).to(torch.int16),
            ((w >> 48) & 0xff).to(torch.uint8), allp[:, 0].contiguous())


def solve_flow(obs_int, entry_int, exit_int, row_ptr, col, cost_int, min_flow, max_flow, group=None, device=None, duals=False):
    """The global flow solve shared between the frame-sharded ranks (hotpath.McfShard / axt_mcf_shard_*): every rank solves
    its own run of time blocks of the (replicated) network on its host threads, ONE all-gather exchanges the runs' states
    (duals and matching, 16 bytes per row and column: ~10 MB for a 300 k-detection timelapse), and every rank joins them
    through the separator rows and finishes. The optimum is unique, so all ranks end with the trajectories of a single-
    process solve, and no broadcast is needed. Returns what hotpath.mcf_solve returns (duals: with the certificate)."""
    import numpy as np
    from . import hotpath as hp
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world & (world - 1):
        return hp.mcf_solve(obs_int, entry_int, exit_int, row_ptr, col, cost_int, min_flow, max_flow, duals=duals)    # not a power of two: replicated
    shard = hp.McfShard(obs_int, entry_int, exit_int, row_ptr, col, cost_int, rank, world)
    on_gpu = dist.get_backend(group) != 'gloo'
    dev = device if on_gpu else torch.device('cpu')
    sizes = torch.zeros((world,), dtype=torch.int64, device=dev)
    sizes[rank] = len(shard.state)
    _collective('flow_state_sizes_allreduce', lambda: dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group))
    sizes = sizes.cpu().numpy()
    longest = max(int(sizes.max()), 1)
    mine = torch.zeros((longest,), dtype=torch.uint8, device=dev)
    mine[:len(shard.state)] = torch.from_numpy(shard.state).to(dev)
    out = torch.empty((world * longest,), dtype=torch.uint8, device=dev)
    _collective('flow_states_allgather', lambda: dist.all_gather_into_tensor(out, mine, group=group))
    out = out.cpu().numpy()
    states = [out[r * longest:r * longest + int(sizes[r])] for r in range(world)]
    return shard.finish(states, min_flow, max_flow, duals=duals)


def assemble_ided_dets_all(blocks, n_frames, reproduce_label_quirk=True):
    """Join the per-rank blocks of IDed_dets_all (AxonDetections._ided_block: rows = identities alive in the rank's
    frames, columns = those frames) into the table of the whole timelapse, exactly as a single process builds it
    (AxonDetections.py:825-842): identities ascending, and -- by default -- the reference's label quirk: frames
    without any IDed detection drop out of the concat, the rest are labelled by position, NaN columns fill the end."""
    import numpy as np
    import pandas as pd
    from .detections import _axon_index, _ided_columns
    blocks = sorted(blocks, key=lambda b: b.columns[0][0] if len(b.columns) else 0)
    ids = sorted({int(name[5:]) for b in blocks for name in b.index})
    row = {i: n for n, i in enumerate(ids)}
    vals = np.full((len(ids), 3 * n_frames), np.nan)
    pos = 0
    for b in blocks:
        v = b.to_numpy()
        rows = np.array([row[int(name[5:])] for name in b.index], np.int64)
        for k in range(v.shape[1] // 3):
            trip = v[:, 3 * k:3 * k + 3]
            if reproduce_label_quirk and np.isnan(trip).all():
                continue                                  # a frame without IDed detections vanishes (:831)
            col = 3 * pos if reproduce_label_quirk else 3 * int(b.columns[3 * k][0])
            if len(rows):
                vals[rows, col:col + 3] = trip
            pos += 1
    return pd.DataFrame(vals, index=_axon_index(np.array(ids, np.int64)), columns=_ided_columns(n_frames), copy=False)


def gather_ided_dets_all(ad, group=None):
    """All ranks: the table of the whole timelapse from every rank's block (one all_gather_object; on demand --
    the hot path itself leaves each rank with its own block)."""
    world = dist.get_world_size(group)
    blocks = [None] * world
    dist.all_gather_object(blocks, ad.IDed_dets_all, group=group)
    return assemble_ided_dets_all(blocks, len(ad), ad.reproduce_label_quirk)
