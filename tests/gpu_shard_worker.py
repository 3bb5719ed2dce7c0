"""Worker of tests/test_gpu_parity.py::test_two_rank_frame_sharding: one rank of a 2-rank (gloo) run in
which both ranks share cuda:0 -- the same code path bench.py --gpus N takes with RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(rank, world, port, q, total_frames, seed):
    sys.path.insert(0, ROOT)
    # (small time blocks, so that the flow solver's tree has leaves to share between the ranks on a 16-frame timelapse)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', AXT_MCF_MIN_LEAF='64',
                      AXT_MCF_THREADS='2')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import axtrack_amd
    from axtrack_amd import synth, params, sharded
    f0, per = sharded.frame_block(total_frames, rank, world)
    frames = synth.synth_frames(total_frames + 4, 512, 512, seed=seed, t_range=(f0, f0 + per + 4))
    model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=per)
    tl = axtrack_amd.Timelapse(frames, name='shard')
    tl_masked = axtrack_amd.Timelapse(frames, name='shard', mask=synth.corridor_mask(512, 512, width=40, pitch=128))
    out = {}
    for mode in ('hungarian', 'mcf', 'mcf+appearance', 'mcf+mask', 'hungarian+mask'):
        P = params.load_parameters()
        P['ASSOCIATION'] = mode.split('+')[0]
        if mode.endswith('appearance'):
            P['MCF_VIS_SIM_WEIGHT'] = 0.2          # the histograms need the pixels: they travel with the detections
        ad = axtrack_amd.AxonDetections(model, tl_masked if mode.endswith('mask') else tl, P, None)
        ad.detect_dataset()
        ad.gather_detections()
        ad.assign_ids()
        block = ad.IDed_dets_all                                  # this rank's frames x the identities alive in them
        assert ad.IDed_dets_block == (f0, f0 + per) and [c[0] for c in block.columns[::3]] == list(range(f0, f0 + per))
        whole = sharded.gather_ided_dets_all(ad)
        out[mode] = (ad.n_ids, ad._track_flat.tobytes(), whole.to_numpy().tobytes(), list(whole.index), block.shape)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def tiles_frames(total_frames, seed):
    """[T,512,1024] timelapse (two tiles) whose right tile is all zero in the frames that rank 0 of 2 reads (its block
    plus halo) and non-zero afterwards: the kept-tile list of the whole timelapse has both tiles, rank 0's own block
    only one."""
    from axtrack_amd import synth
    frames = synth.synth_frames(total_frames + 4, 512, 1024, seed=seed)
    frames[:total_frames // 2 + 4, :, 512:] = 0
    return frames


def run_tiles(rank, world, port, q, total_frames, seed):
    """Two tiles, one of them empty in rank 0's frames (the advisor's case): the tile list must be the timelapse-wide
    one on both ranks (Timelapse.sync_tile_occupancy), or the ranks bring different array shapes to the all-gather."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import axtrack_amd
    from axtrack_amd import synth, params, sharded
    f0, per = sharded.frame_block(total_frames, rank, world)
    frames = tiles_frames(total_frames, seed)[f0:f0 + per + 4]
    model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=2 * per)
    tl = axtrack_amd.Timelapse(frames, name='tiles')
    local = list(tl.tile_yx)
    tl._tile_yx = None
    P = params.load_parameters()
    P['ASSOCIATION'] = 'hungarian'
    # without the sync the ranks disagree on the capacity and the gather must refuse
    refused = False
    ad = axtrack_amd.AxonDetections(model, tl, P, None)
    ad.detect_dataset()
    try:
        ad.gather_detections()
    except ValueError:
        refused = True
    tiles = tl.sync_tile_occupancy()
    ad = axtrack_amd.AxonDetections(model, tl, P, None)
    ad.detect_dataset()
    ad.gather_detections()
    ad.assign_ids()
    whole = sharded.gather_ided_dets_all(ad)
    q.put((rank, dict(local=local, tiles=tiles, refused=refused, n_ids=ad.n_ids, track=ad._track_flat.tobytes(),
                      table=whole.to_numpy().tobytes(), cap=int(ad.d_conf.shape[1]))))
    dist.barrier()
    dist.destroy_process_group()
