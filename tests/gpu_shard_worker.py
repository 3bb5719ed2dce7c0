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
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
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
