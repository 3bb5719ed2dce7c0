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
