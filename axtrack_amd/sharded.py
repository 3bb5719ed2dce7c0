"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed).

Detection is embarrassingly parallel over frames: rank r owns a contiguous block of detection
frames and reads a +-2-frame input halo; no communication. Association is global, so the only
exchange on the path is ONE all-gather of the per-frame detection lists (a few MB: latency-bound;
RCCL over xGMI on GPUs, gloo on CPU in the tests). Every rank then holds all detections and runs
the same deterministic integer-cost solve (replicated: no broadcast needed).
"""
import torch
import torch.distributed as dist


def frame_block(n_frames_total, rank, world):
    """Contiguous, equal blocks (the caller pads n_frames_total to a multiple of world)."""
    if n_frames_total % world:
        raise ValueError(f'{n_frames_total} detection frames do not split evenly over {world} ranks')
    per = n_frames_total // world
    return rank * per, per


def all_gather_detections(conf, x, y, count, group=None):
    """conf f32 [F,cap], x/y i32 [F,cap], count i32 [F] per rank -> the same arrays for all
    world*F frames, rank-major. One collective on a packed i32 buffer [F, 3*cap+1]."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return conf, x, y, count
    world = dist.get_world_size(group)
    F, cap = conf.shape
    packed = torch.cat([conf.contiguous().view(torch.int32), x, y, count.view(F, 1)], dim=1).contiguous()
    out = torch.empty((world * F, 3 * cap + 1), dtype=torch.int32, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    g_conf = out[:, :cap].contiguous().view(torch.float32)
    g_x = out[:, cap:2 * cap].contiguous()
    g_y = out[:, 2 * cap:3 * cap].contiguous()
    g_count = out[:, 3 * cap].contiguous()
    return g_conf, g_x, g_y, g_count


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
