#!/usr/bin/env python3
"""bench.py -- frames/sec of AxTrack's detect + associate hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (axtrack_amd.inference: CNN forward -> decode/stitch/NMS ->
[all-gather] -> association -> IDed_dets_all; the list of non-empty tiles belongs to the dataset, as in the
reference, and is computed once when the timelapse is first used)
over a synthetic 512x512 timelapse that is already resident in HBM. Default workload is BASELINE
config "c3" (512x512x256, detection + association); "c2" is detection only; "c4" / "c5" are BASELINE configs 4 and 5
(1024x1024 frames, the reference's global min-cost-flow tracker; c5 under the corridor mask with path costs on the masked
grid): frame-sharded like c3, `--gpus 1` runs ONE GPU's share of the 8-GPU configuration (132 / 68 input frames).

Multi-GPU (weak scaling): the timelapse has N x 252 detection frames, rank r detects its own
contiguous block (reading a 2-frame halo), ONE all-gather of the detection lists over RCCL, then the
association (Hungarian: every rank its own frame pairs + one MAX all-reduce of the links; min-cost flow: the closed-form
arcs of an all-ones mask are rebuilt on every rank, so the all-gather stays the only collective before the replicated solve), and
every rank materialises its own block of IDed_dets_all (its frames x the identities alive in them).

Prints ONE JSON line on rank 0 (contract: see the task description), with
  roofline     -- dominant kernel: algorithmic FLOPs / HIP-event time measured inside the timed
                  region on the launch stream, against the f32 MFMA peak (157.3 TFLOP/s);
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm, oracle/) timed on this
                  box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2516.6    # dense bf16 MFMA (16 x the f32 rate); the bf16x3 line prices a multiply-add at 6 products


# BASELINE configs 4 and 5: (frame size, input frames of the whole configuration, GPUs it is stated for)
BIG_CONFIGS = {'c4': (1024, 1024, 8), 'c5': (1024, 512, 8)}


def resolve_workload(args):
    """Fill in what the workload's name implies: frame size, input frames per GPU, association. c4 / c5 are stated for 8 GPUs:
    a rank gets one eighth of the configuration's detection frames (rounded up) whatever --gpus is, so that --gpus 8 is the
    configuration itself and --gpus 1 one GPU's share of it."""
    wl = args.workload
    if wl in BIG_CONFIGS:
        size, t_all, gpus = BIG_CONFIGS[wl]
        if args.size is None:
            args.size = size
        if args.frames is None:
            args.frames = -(-(t_all - 4) // gpus) + 4
        args.assoc = 'mcf'                      # the configurations name the global flow tracker
        args.input = 'hbm'
    else:
        if args.size is None:
            args.size = 512
        if args.frames is None:
            args.frames = 256
    args.associates = wl in ('c3', 'c4', 'c5')
    args.big = wl in BIG_CONFIGS
    return args


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='c3', choices=['c2', 'c3', 'c4', 'c5', 'assoc-c3', 'assoc-c4'],
                    help="c3 (default, BASELINE config 3) / c2 (detection only); c4 / c5: BASELINE configs 4 and 5 (1024x1024 frames, global "
                         "min-cost flow; c5 with synth.corridor_mask and path costs on the masked grid), per GPU one eighth of the "
                         "configuration's frames (--frames 132 / 68 unless given); assoc-c3 / assoc-c4: ASSOCIATION ONLY on the detections of a "
                         "scene of moving growth cones (births, deaths, misses, clutter; synth.synth_detections) at the size of config 3 / 4 -- "
                         "separate lines for the flow tracker, whose cost depends on the scene (the random-init detector gives a static one)")
    ap.add_argument('--assoc', default='hungarian', choices=['hungarian', 'mcf'],
                    help="association of workload c3: 'hungarian' = BASELINE config 3 as written (frame-to-frame), "
                         "'mcf' = the reference's global min-cost-flow tracker")
    ap.add_argument('--arith', default='f32', choices=['f32', 'f32_winograd', 'f32_direct', 'bf16x3'],
                    help="arithmetic of the stride-1 conv blocks: 'f32' (default, the headline; = 'f32_winograd': Winograd "
                         "F(2x2,3x3) on the f32 matrix pipe, every operation f32), 'f32_direct' (direct convolution on the f32 "
                         "matrix pipe) or the opt-in 'bf16x3' (three bf16 terms per operand on the bf16 matrix pipe, f32 "
                         "accumulation): SEPARATE lines")
    ap.add_argument('--input', default='hbm', choices=['hbm', 'host'],
                    help="'hbm' (the headline): preprocessed f32 frames resident in HBM when the timed region starts; 'host': raw "
                         "uint16 frames in pinned host memory -- every pass copies them in chunks on a second stream beside the "
                         "preprocessing and the CNN of the previous chunk (a SEPARATE line: the PCIe-inclusive rate)")
    ap.add_argument('--no-host-variant', action='store_true', help='skip the host-resident-input passes after the timed region')
    ap.add_argument('--chunk', type=int, default=96, help='largest chunk of input frames of --input host (the chunks grow from 16)')
    ap.add_argument('--frames', type=int, default=None, help='input frames per GPU (T_all): 256 (c2, c3), 132 (c4), 68 (c5)')
    ap.add_argument('--size', type=int, default=None, help='frame height = width: 512 (c2, c3), 1024 (c4, c5)')
    ap.add_argument('--cpu-frames', type=int, default=252, help='detection frames of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--no-profile', action='store_true', help='do not bracket kernels with HIP events')
    ap.add_argument('--no-verify', action='store_true',
                    help='skip the self-check of the benchmarked configuration against the CPU oracle (outside the timed region)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='gloo only for rehearsals')
    ap.add_argument('--single-device', action='store_true',
                    help='rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)')
    ap.add_argument('--replicated-solve', action='store_true', help='N > 1, --assoc mcf: every rank solves the whole flow problem (round 2) instead of sharing it')
    ap.add_argument('--launch-check', action='store_true',
                    help='no GPU work: every rank joins a gloo group, one all-reduce, rank 0 prints what it saw (tests the launcher)')
    args = ap.parse_args()
    resolve_workload(args)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # a bare `python bench.py --gpus N`: this process becomes the launcher of N ranks (it never touches the GPU)
        raise SystemExit(launch_ranks(args.gpus))
    global torch
    import torch
    if args.workload.startswith('assoc'):
        return assoc_only(args)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    if args.launch_check:
        import torch.distributed as dist
        dist.init_process_group('gloo')
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({'launch_check': True, 'world': dist.get_world_size(), 'rank_sum': int(t.item()),
                              'master': f"{os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}"}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')

    import axtrack_amd
    from axtrack_amd import synth, params

    H = W = args.size
    per_rank = args.frames - 4                      # detection frames per GPU
    total_frames = per_rank * world
    T_all_global = total_frames + 4
    # this rank's block of the global timelapse, with its 2-frame halo on both sides
    f0 = rank * per_rank
    frames = synth.synth_frames(T_all_global, H, W, seed=0, t_range=(f0, f0 + per_rank + 4))
    mask = None
    if args.workload == 'c5':                       # BASELINE config 5: occlusion mask, path costs on the masked grid
        mask = synth.corridor_mask(H, W, width=40, pitch=128)
        frames *= mask[None].astype(np.float32)
    frames_host = frames if (rank == 0 and not args.no_verify) else None     # the checker's copy (rank 0's block)
    sd = synth.synth_state_dict(42)
    P = params.load_parameters()
    P['DEVICE'] = str(dev)
    P['ASSOCIATION'] = args.assoc
    P['CNN_ARITH'] = args.arith
    P['MCF_SHARDED_SOLVE'] = not args.replicated_solve
    winograd = args.arith in ('f32', 'f32_winograd')
    wino_blocks = ('conv2', 'conv4', 'conv5', 'conv7', 'conv8', 'conv10')  # the stride-1 blocks (Winograd in the default arithmetic)
    n_tiles = (-(-H // 512)) * (-(-W // 512))
    model = axtrack_amd.Detector(sd, max_batch=min(per_rank * n_tiles, 1024), device=dev)
    raw_pinned = None
    if args.input == 'host':
        if world > 1:
            raise SystemExit('--input host is a one-GPU line')
        # the raw 16-bit timelapse whose preprocessing gives (nearly) these frames: offset 121, clip 55, log2, train-set std
        scale = params.DEPLOYED_STND_SCALER[1][0]
        raw = np.clip((2.0 ** (frames.astype(np.float64) * scale) - 1.0) * 65535.0 + 121.0 * (frames > 0), 0, 65535).astype(np.uint16)
        raw_pinned = torch.from_numpy(raw.view(np.int16)).pin_memory()
        host_tl = lambda: axtrack_amd.Timelapse.from_host_u16(raw_pinned, name='bench', offset=121, clip=55, scale=scale,
                                                             chunk_frames=args.chunk, device=dev)
        tl = host_tl()
        for _ in tl.stream_chunks():
            pass
        if frames_host is not None:
            frames_host = tl.frames.cpu().numpy()          # the checker reads what the preprocessing produced
    else:
        tl = axtrack_amd.Timelapse(frames, name='bench', mask=mask, device=dev)
    del frames
    if world > 1:
        tl.sync_tile_occupancy()          # the kept-tile list is a property of the whole timelapse (Timelapse.py:551-558)

    def step():
        ad = axtrack_amd.AxonDetections(model, host_tl() if raw_pinned is not None else tl, P, None)
        ad.detect_dataset(cache=None)
        if args.associates:
            if world > 1:
                ad.gather_detections()
            ad.assign_ids(None, None)
        return ad

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        ad = step()
    sync_all()
    # Per-kernel table: one extra untimed pass with every launch bracketed by HIP events (that costs ~8 % of the
    # pass, so it stays outside the timed region). Inside the timed region only the dominant kernel is bracketed:
    # its time there is what roofline.achieved is computed from.
    table, dom_idx = None, None
    if not args.no_profile:
        model.set_profiling(True)
        model.read_profile()
        step()
        sync_all()
        table = model.read_profile()
        dom_idx = max(range(len(table)), key=lambda i: table[i]['ms'])
        model.set_profiling(True, only=dom_idx)
        model.read_profile()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ad = step()
    sync_all()
    dt = time.perf_counter() - t0
    prof = None
    if not args.no_profile:
        prof = model.read_profile()
        model.set_profiling(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- stage breakdown (one extra, untimed, synchronised step on rank 0's clock)
    stages = {}
    def timed(name, fn):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(dev)
        stages[name] = round((time.perf_counter() - t) * 1e3, 3)
        return r
    from axtrack_amd import sharded
    ad2 = axtrack_amd.AxonDetections(model, host_tl() if raw_pinned is not None else tl, P, None)
    timed('detect_ms', lambda: ad2.detect_dataset(cache=None))
    if args.associates:
        sharded.COLLECTIVE_MS = {}                     # per-collective wall time of this (untimed, synchronised) step
        if world > 1:
            timed('allgather_ms', ad2.gather_detections)
        timed('associate_ms', lambda: ad2.assign_ids(None, None))
        collectives = {k: round(v, 3) for k, v in sharded.COLLECTIVE_MS.items()}
        sharded.COLLECTIVE_MS = None

    # every rank must hold the same association: compare a hash of the trajectory ids of the whole timelapse
    same_on_all_ranks = None
    if args.associates and world > 1:
        import hashlib
        digest = hashlib.sha256(np.ascontiguousarray(ad._track_flat).tobytes()).digest()[:8]
        h = torch.tensor([int.from_bytes(digest, 'little', signed=True)], dtype=torch.int64, device=dev)
        lo, hi = h.clone(), h.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same_on_all_ranks = bool(lo.item() == hi.item())
        assert same_on_all_ranks, 'ranks disagree on the trajectories of the gathered timelapse'

    # the direct-convolution f32 kernels on the same workload, measured in the same run (after the timed region)
    direct = None
    if winograd and world == 1 and not args.big:
        Pd = dict(P, CNN_ARITH='f32_direct')
        def step_direct():
            a = axtrack_amd.AxonDetections(model, tl, Pd, None)
            a.detect_dataset(cache=None)
            if args.workload == 'c3':
                a.assign_ids(None, None)
        step_direct()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(3):
            step_direct()
        torch.cuda.synchronize(dev)
        dtd = (time.perf_counter() - t) / 3
        direct = {'cnn_arith': 'f32_direct', 'value': round(total_frames / dtd, 2), 'unit': 'frames/s',
                  'ms_per_step': round(dtd * 1e3, 3), 'steps': 3}
        model.set_arith(args.arith)

    # the same pass from host-resident raw input, measured in the same run (after the timed region): the PCIe-inclusive rate
    host_variant = None
    if args.input == 'hbm' and world == 1 and not args.no_host_variant and not args.big:
        scale = params.DEPLOYED_STND_SCALER[1][0]
        fr = tl.frames.cpu().numpy()
        raw = np.clip((2.0 ** (fr.astype(np.float64) * scale) - 1.0) * 65535.0 + 121.0 * (fr > 0), 0, 65535).astype(np.uint16)
        del fr
        rp = torch.from_numpy(raw.view(np.int16)).pin_memory()
        def step_host():
            t_ = axtrack_amd.Timelapse.from_host_u16(rp, name='bench', offset=121, clip=55, scale=scale, chunk_frames=args.chunk, device=dev)
            a = axtrack_amd.AxonDetections(model, t_, P, None)
            a.detect_dataset(cache=None)
            if args.workload == 'c3':
                a.assign_ids(None, None)
            return a
        for _ in range(3):
            step_host()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(5):
            ah = step_host()
        torch.cuda.synchronize(dev)
        dth = (time.perf_counter() - t) / 5
        host_variant = {'input': 'host_u16', 'value': round(total_frames / dth, 2), 'unit': 'frames/s', 'ms_per_step': round(dth * 1e3, 3),
                        'steps': 5, 'chunk_frames': args.chunk, 'h2d_bytes_per_step': int(rp.numel() * 2),
                        'detections': int(ah._host_dets()[0].sum()),
                        'what': 'raw uint16 frames in pinned host memory, copied in chunks on a second stream beside axt_preprocess_u16 '
                                'and the CNN of the previous chunk; python bench.py --input host prints it as a line of its own'}
        del rp, raw

    # the other association variant, measured in the same run (untimed region, one step) for transparency
    other = None
    if args.workload == 'c3' and world == 1:
        P2 = dict(P, ASSOCIATION='mcf' if args.assoc == 'hungarian' else 'hungarian')
        ad3 = axtrack_amd.AxonDetections(model, tl, P2, None)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        ad3.detect_dataset(cache=None)
        ad3.assign_ids(None, None)
        torch.cuda.synchronize(dev)
        dt3 = time.perf_counter() - t
        other = {'association': P2['ASSOCIATION'], 'value': round(total_frames / dt3, 2), 'unit': 'frames/s',
                 'ms_per_step': round(dt3 * 1e3, 3), 'n_ids': ad3.n_ids, 'steps': 1}

    # A real timelapse is processed once: the pass over a FRESH timelapse object of the same frames (the kept-tile list is
    # computed again, the identity count that sizes IDed_dets_all is fetched instead of guessed from the previous pass),
    # timed after the timed region for the steady_state note of the line
    fresh_ms = fresh_next_ms = None
    if world == 1 and args.input == 'hbm':
        from axtrack_amd import detections as _det
        fresh_all = []
        for _ in range(6 if not args.big else 2):
            _det._IDS_GUESS.clear()
            tl_fresh = axtrack_amd.Timelapse(tl.frames, name='bench', mask=mask, device=dev)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            a = axtrack_amd.AxonDetections(model, tl_fresh, P, None)
            a.detect_dataset(cache=None)
            if args.associates:
                a.assign_ids(None, None)
            torch.cuda.synchronize(dev)
            fresh_all.append((time.perf_counter() - t) * 1e3)
            del a, tl_fresh
        fresh_ms = round(fresh_all[0], 3)
        fresh_next_ms = round(sum(fresh_all[1:]) / len(fresh_all[1:]), 3)

    # the flow tracker's optimality certificate (axt_mcf_solve_duals): one more untimed step on every rank (the shared solve is
    # collective), checked on rank 0 in verify()
    cert = None
    if args.associates and args.assoc == 'mcf' and not args.no_verify:
        Pc = dict(P, MCF_CERTIFICATE=True)
        ac = axtrack_amd.AxonDetections(model, tl, Pc, None)
        ac.detect_dataset(cache=None)
        if world > 1:
            ac.gather_detections()
        ac.assign_ids(None, None)
        cert = ac.mcf_certificate
        assert np.array_equal(ac._track_flat, ad._track_flat), 'the certified step found other trajectories than the timed one'
        del ac

    if rank == 0:
        value = total_frames * args.steps / dt
        out = {
            'metric': (f'frames/sec end-to-end detect+associate, {H}x{W}xT timelapse' if args.associates
                       else 'frames/sec detection only (CNN forward + NMS), 512x512xT timelapse')
                      + (' -- raw uint16 input in host memory, H2D copy and preprocessing inside every pass (PCIe-inclusive)' if args.input == 'host' else ''),
            'value': round(value, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32' if args.arith != 'bf16x3' else 'f32 via 3 x bf16 split operands (6 partial products, f32 accumulate) in conv blocks 2-8; f32 elsewhere',
            'data': 'synthetic',
            'config': {'workload': f'{args.workload}: synthetic {H}x{W}x{args.frames} grayscale timelapse per GPU, '
                                   + (f'detection + path-cost matrix + {"Hungarian (frame-to-frame)" if args.assoc == "hungarian" else "global min-cost-flow"} association (IDed_dets_all)'
                                      if args.associates else 'detection only')
                                   + ('' if not args.big else
                                      f'; BASELINE config {args.workload[1]} = {BIG_CONFIGS[args.workload][0]}x{BIG_CONFIGS[args.workload][0]}x{BIG_CONFIGS[args.workload][1]} on '
                                      f'{BIG_CONFIGS[args.workload][2]} GPUs' + (', occlusion mask (40-px corridors on a 128-px lattice), path costs on the masked grid' if mask is not None else ', all-ones mask')
                                      + f': this run is {world} of {BIG_CONFIGS[args.workload][2]} GPU shares ({per_rank} detection frames per GPU), the timelapse it solves has {total_frames} detection frames'),
                       'association': args.assoc if args.associates else None,
                       'detection_frames_per_gpu': per_rank, 'tiles_per_frame': n_tiles,
                       'weights': 'random-init (synth seed 42)', 'parallelism': f'frame-sharded x{world}',
                       'input': 'hbm_resident' if args.input == 'hbm' else 'host_u16',
                       'input_detail': ('preprocessed f32 frames resident in HBM when the timed region starts' if args.input == 'hbm' else
                                        f'raw uint16 frames in pinned host memory: every pass copies them on a second stream (16-frame pieces) beside '
                                        f'axt_preprocess_u16 and the CNN of the frames already there (chunks of 16, 32, 48, 64, 80, then {args.chunk} frames; PCIe-inclusive)'),
                       'steady_state': {'what': 'the timed passes run over ONE timelapse object: its kept-tile list (a property of the dataset, Timelapse.py:551-558) '
                                                'is computed by the first pass and kept, and IDed_dets_all is sized from the identity count of the previous pass '
                                                'over a timelapse of this shape instead of a device-to-host round trip; packed weights belong to the Detector',
                                        'fresh_timelapse_pass_ms': fresh_ms,
                                        'fresh_timelapse_next_passes_ms': fresh_next_ms,
                                        'fresh_timelapse_pass': 'passes over NEW Timelapse objects of the same frames with the identity-count guess cleared (after the timed region): the first one, and the mean of the following ones (each again over a new object): the first also pays what a process pays once (allocator pools for the new buffers)'},
                       'cnn_arith': args.arith,
                       'conv_algorithm': ('Winograd F(2x2,3x3), f32, for the six stride-1 conv blocks (2,4,5,7,8,10); direct for the two stride-2 blocks'
                                          if winograd else 'direct') + ('; the two stride-2 blocks fused into one kernel' if getattr(model, 'fused_front', False) else '')},
            'stages': stages,
            'detections': int(ad._host_dets()[0].sum()),
        }
        if world > 1:
            out['rccl_world'] = dist.get_world_size() if args.backend == 'nccl' else 0
            out['backend'] = args.backend
            if args.associates:
                out['collectives_ms'] = collectives
                out['tracks_identical_on_all_ranks'] = same_on_all_ranks
        if not args.no_verify:
            out.update(verify(args, ad, frames_host, sd, per_rank, world, mask, cert))
        if args.associates:
            out['n_ids'] = getattr(ad, 'n_ids', None)
            if other:
                out['other_association_variant'] = other
        if direct:
            out['other_arithmetic_variant'] = direct
        if host_variant:
            out['host_input_variant'] = host_variant
        if prof:
            # FLOPs a kernel EXECUTES on the matrix pipe: a Winograd block multiplies 16 times per 2x2 output tile, input and
            # output channel where the direct convolution multiplies 36 times (its transforms are additions on the vector pipe)
            def executed(k):
                return k['flops_per_tile'] * k['tiles'] * (16.0 / 36.0 if winograd and k['name'].split()[0] in wino_blocks else 1.0)
            dom = prof[dom_idx]                                   # bracketed inside the timed region
            flops = executed(dom)
            achieved = flops / (dom['ms'] * 1e-3) / 1e12
            cnn_ms = sum(k['ms'] for k in table)                  # one untimed pass, all launches bracketed
            cnn_flops = sum(executed(k) for k in table if 'reduce' not in k['name'])
            cnn_flops_direct = sum(k['flops_per_tile'] * k['tiles'] for k in table if 'reduce' not in k['name'])
            peak = PEAK_F32_MFMA_TFLOPS
            if args.arith == 'bf16x3' and dom['name'].split()[0] in ('conv2', 'conv4', 'conv5', 'conv7', 'conv8'):
                peak = PEAK_BF16_MFMA_TFLOPS / 6.0      # algorithmic f32 multiply-adds cost six bf16 products each
            out['roofline'] = {
                'bound': 'mfma', 'kernel': dom['name'], 'achieved': round(achieved, 2), 'peak': round(peak, 1),
                'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                **committed_counters(table, dom['name'], winograd, wino_blocks, cnn_ms, args),
                'avg_launch_ms': round(dom['ms'] / max(dom['launches'], 1), 4),
                'flops_per_launch': flops / max(dom['launches'], 1),
                'flops': ('executed on the matrix pipe (Winograd: 16/36 of the direct convolution\'s); '
                          'direct_equivalent = the same launches priced as direct convolutions' if winograd else 'direct convolution'),
                'direct_equivalent': round(dom['flops_per_tile'] * dom['tiles'] / (dom['ms'] * 1e-3) / 1e12, 2),
                # SURVEY.md 8d prices the layer as a direct convolution (2 x Hout^2 x Cout x Cin x 9 per tile): that count over
                # the same time, next to the executed one
                'algorithmic': {'achieved': round(dom['flops_per_tile'] * dom['tiles'] / (dom['ms'] * 1e-3) / 1e12, 2),
                                'frac': round(dom['flops_per_tile'] * dom['tiles'] / (dom['ms'] * 1e-3) / 1e12 / peak, 4),
                                'flops_per_launch': dom['flops_per_tile'] * dom['tiles'] / max(dom['launches'], 1),
                                'whole_cnn_frac': round(cnn_flops_direct / (cnn_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)},
                'measured': 'HIP events around every launch of this kernel inside the timed region',
                'whole_cnn': {'achieved': round(cnn_flops / (cnn_ms * 1e-3) / 1e12, 2),
                              'frac': round(cnn_flops / (cnn_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                              'direct_equivalent': round(cnn_flops_direct / (cnn_ms * 1e-3) / 1e12, 2),
                              'ms_per_step': round(cnn_ms, 3),
                              'measured': 'one untimed pass with every launch bracketed'},
                'kernels': [{'name': k['name'], 'ms_per_step': round(k['ms'], 4),
                             'tflops': round(executed(k) / max(k['ms'], 1e-9) / 1e9, 2)}
                            for k in table if k['launches']],
            }
        if world == 1 and args.cpu_frames > 0:
            out['cpu_baseline'] = cpu_baseline(args, sd, synth)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def assoc_only(args):
    """Association-only lines: the detections of a scene of MOVING growth cones (synth.synth_detections: ~n_alive cones
    per frame that are born, move <= 10 px per frame, are missed 8 % of the time and die; 8 % clutter) are resident in HBM;
    one step = AxonDetections.assign_ids(): observation costs, admissible arcs (GPU), the association (--assoc mcf: the
    reference's global min-cost-flow tracker, host solve; hungarian: the frame-to-frame variant, GPU) and IDed_dets_all.
    Verified outside the timed region: the flow tracker against the successive-shortest-path solver (same trajectories
    and cost), the frame-to-frame variant against the oracle's."""
    import axtrack_amd
    from axtrack_amd import synth, params, hotpath as hp
    from axtrack_amd.detections import _cost_units_on_device
    F, size, alive = {'assoc-c3': (252, 512, 90), 'assoc-c4': (1020, 1024, 380)}[args.workload]
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    d = synth.synth_detections(F, size, size, n_alive=alive, seed=0)
    P = params.load_parameters()
    P['ASSOCIATION'] = args.assoc
    P['MCF_MAX_FLOW'] = 100000                      # (the deployed 450 is for the example's field of view; config 4 holds more cones)
    tl = axtrack_amd.Timelapse(torch.zeros((5, size, size)), name='assoc', device=dev)
    cd, cx, cy, cc = (torch.from_numpy(d[k]).to(dev) for k in ('conf', 'x', 'y', 'count'))

    def step():
        ad = axtrack_amd.AxonDetections(None, tl, P, None)
        ad.set_detections(cd, cx, cy, cc)
        ad.assign_ids(None, None)
        return ad
    for _ in range(args.warmup):
        ad = step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ad = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / args.steps
    # the arc builder alone (HBM-bound integer work), with events on its stream
    dmax, units = _cost_units_on_device(P, 500, dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hp.build_arcs(cx, cy, cc, size, size, dmax, units)
    e0.record()
    for _ in range(5):
        arcs = hp.build_arcs(cx, cy, cc, size, size, dmax, units)
    e1.record()
    torch.cuda.synchronize(dev)
    arcs_ms = e0.elapsed_time(e1) / 5
    n_det, n_arcs = int(d['count'].sum()), int(arcs[1].numel())
    arc_bytes = 12 * n_det * (1 + len(dmax)) + 15 * n_arcs        # anchors read per (frame, gap), 15 B per admitted arc
    ok, what = None, 'not checked'
    if not args.no_verify:
        if args.assoc == 'mcf':
            os.environ['AXT_MCF_FORCE_SSP'] = '1'
            ref = step()
            del os.environ['AXT_MCF_FORCE_SSP']
            ok = bool(np.array_equal(ref._track_flat, ad._track_flat) and ref.mcf_total_cost == ad.mcf_total_cost)
            what = 'trajectories and total cost equal to the successive-shortest-path solver on the same network'
        else:
            from oracle import oracle as orc
            dets = [(d['conf'][t, :d['count'][t]], d['x'][t, :d['count'][t]].astype(np.int64), d['y'][t, :d['count'][t]].astype(np.int64)) for t in range(F)]
            trajs = orc.hungarian_assoc(dets, size, size, dict(orc.DEFAULTS))
            frame_of = np.repeat(np.arange(F), d['count'])
            offs = np.concatenate([[0], np.cumsum(d['count'])])
            got = {}
            for k, tid in enumerate(ad._track_flat):
                if tid >= 0:
                    got.setdefault(int(tid), []).append((int(frame_of[k]), int(k - offs[frame_of[k]])))
            ok = [sorted(got[i]) for i in sorted(got)] == trajs
            what = 'trajectories equal to the oracle (hungarian)'
    # how well the association recovers the scene's cones: detections of one cone that share one identity
    truth = np.concatenate([d['truth'][t, :d['count'][t]] for t in range(F)])
    tr = ad._track_flat
    both = (truth >= 0) & (tr >= 0)
    purity = None
    if both.any():
        pairs = np.stack([truth[both], tr[both]], 1)
        _, inv, cnts = np.unique(pairs, axis=0, return_inverse=True, return_counts=True)
        best = {}
        for (c_, _t), n_ in zip(np.unique(pairs, axis=0), cnts):
            best[c_] = max(best.get(c_, 0), n_)
        purity = round(sum(best.values()) / max(int((truth >= 0).sum()), 1), 4)
    out = {'metric': f'frames/sec association only ({"global min-cost flow" if args.assoc == "mcf" else "frame-to-frame Hungarian"}), '
                     f'{size}x{size}x{F + 4} scene of moving growth cones',
           'value': round(F / dt, 2), 'unit': 'frames/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': round(dt * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'int64 (integer arc costs); f64 observation costs', 'data': 'synthetic',
           'config': {'workload': f'{args.workload}: association only -- {F} frames of {size}x{size}, ~{alive} moving cones per frame (born, '
                                  f'<= 10 px per frame, missed 8 %, dying; 8 % clutter), detections resident in HBM',
                      'association': args.assoc, 'input': 'detections_hbm_resident', 'detections': n_det, 'arcs': n_arcs},
           'n_ids': ad.n_ids, 'cones_in_scene': int(len(np.unique(truth[truth >= 0]))), 'detections_of_a_cone_under_its_main_identity': purity,
           'verified': ok, 'verify': what,
           'roofline': {'bound': 'hbm', 'kernel': 'axt_build_arcs (count + fill passes)', 'achieved': round(arc_bytes / (arcs_ms * 1e-3) / 1e9, 2),
                        'peak': 8000.0, 'unit': 'GB/s', 'frac': round(arc_bytes / (arcs_ms * 1e-3) / 1e9 / 8000.0, 5), 'traffic': None,
                        'avg_launch_ms': round(arcs_ms, 4), 'bytes_per_launch': arc_bytes,
                        'note': 'algorithmic bytes (12 B per detection and (frame, gap) + 15 B per admitted arc) over HIP-event time; the pass is '
                                'latency-bound at this size; the flow solve itself is host work and has no roofline'},
           'solver_threads': min(len(os.sched_getaffinity(0)), 16)}
    print(json.dumps(out), flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks as CHILD processes of this one
    (same command line; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, exactly what
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` sets), pass rank 0's JSON line through and return
    non-zero if any rank fails. The parent initialises neither torch nor the GPU, and nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is passed through by a reader thread while ALL children are polled: a rank that dies leaves the others
    # waiting in a collective (until the backend's timeout, tens of minutes), so the first failure ends the run at once
    import threading
    def pump():
        for line in procs[0].stdout:                    # rank 0 prints the one JSON line; library chatter goes to stderr
            out = sys.stdout if line.lstrip().startswith(b'{') else sys.stderr
            out.buffer.write(line)
            out.flush()
    reader = threading.Thread(target=pump, daemon=True)
    reader.start()
    rc = 0
    try:
        alive = set(range(n))
        while alive and rc == 0:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    print(f'bench.py: rank {r} exited with code {code}', file=sys.stderr)
                    rc = code if code > 0 else 1
                    break
            if alive and rc == 0:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        reader.join(timeout=5)
    return rc


def verify(args, ad, frames_host, sd, per_rank, world, mask=None, cert=None):
    """Self-check of the configuration that was just timed (same Detector, same launch shapes), outside the timed region,
    against the CPU oracle: YOLO grids of 8 frames sampled across every front-layer launch within 1e-5 of the oracle's
    f32 forward pass; the detection lists of ALL of rank 0's frames bit-exact given those grids; the trajectories equal
    to the oracle's association (one GPU; with several ranks the association is checked through the cross-rank hash).
    The flow tracker's oracle is a Bellman-Ford solver (~40 s at config 3's size)."""
    from oracle import oracle as orc
    t0 = time.perf_counter()
    yolo = ad._yolo.cpu().numpy()                          # this rank's frames [per_rank, n_tiles, 12, 12, 3]
    F, n_tiles = yolo.shape[:2]
    per_launch = max(128 // n_tiles, 1)
    sample = sorted({0, min(per_launch - 1, F - 1), per_launch % F, (per_launch + 1) % F, F // 3, F // 2, (2 * F) // 3, F - 1})
    err = 0.0
    for t in sample:
        ref = orc.cnn_forward(sd, orc.frame_tile_stack(frames_host, t, ad.tile_yx))
        err = max(err, float(np.abs(yolo[t] - ref).max() / (1.0 + np.abs(ref).max())))
    ok_cnn = err < 1e-5                                    # 10x what the kernels deliver (4e-7 ... 1.5e-6)
    ref_dets = orc.detect_from_yolo(list(yolo), ad.tile_yx)
    cnt, conf, x, y = ad._host_dets()
    ok_det = True
    for t, (rc, rx, ry) in enumerate(ref_dets):            # after a gather the arrays hold all ranks' frames; rank 0's come first
        n = len(rc)
        ok_det &= bool(cnt[t] == n and np.array_equal(conf[t, :n], rc) and np.array_equal(x[t, :n], rx)
                       and np.array_equal(y[t, :n], ry))
    ok_assoc, what = None, 'not checked'
    extra = {}
    if cert is not None:
        # optimality of the flow tracker's trajectories: complementary slackness of its node potentials over all arcs
        from tests.helpers import check_flow_certificate
        try:
            proof = check_flow_certificate(cert['obs'], cert['entry'], cert['exit'], cert['row_ptr'], cert['col'], cert['cost'], cert['next'],
                                           cert['track'], cert['total_cost'], cert['potentials'], cert['min_flow'], cert['max_flow'])
            extra['flow_certificate'] = dict(proof, ok=True, what='reduced cost >= 0 on every arc without flow, <= 0 on every arc with flow '
                                                                  '(axt_mcf_solve_duals), flow feasible, total cost recomputed')
        except AssertionError as e:
            extra['flow_certificate'] = {'ok': False, 'why': str(e)}
    if args.big:
        # configs 4 / 5: the oracle's Bellman-Ford tracker does not finish at this size; the certificate above proves the
        # optimum, and sampled rows of the network are compared with the oracle's path lengths, admission and integer costs
        Pc = dict(orc.DEFAULTS)
        offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        det = lambda t: (conf[t, :cnt[t]], x[t, :cnt[t]].astype(np.int64), y[t, :cnt[t]].astype(np.int64))
        rng = np.random.default_rng(5)
        H, W = frames_host.shape[1:]
        n_rows = n_arcs = 0
        ok_arcs = cert is not None
        if cert is not None:
            row_ptr, col, cost = cert['row_ptr'], cert['col'], cert['cost']
            for t in (0, F // 2):
                pick = np.arange(cnt[t]) if mask is None else np.sort(rng.choice(cnt[t], min(4, cnt[t]), replace=False))
                src = tuple(a[pick] for a in det(t))
                want = {int(i): [] for i in pick}
                for g in (1, 2):
                    if t + g >= len(cnt):
                        continue
                    D = orc.path_matrix(src, det(t + g), H, W, mask)
                    c = orc.transition_cost(D, g, Pc['MCF_MISS_RATE'])
                    for r, j in zip(*np.nonzero(c < Pc['MCF_EDGE_COST_THR'])):
                        a, b = int(offs[t] + pick[r]), int(offs[t + g] + j)
                        want[int(pick[r])].append((b, orc.arc_cost_int(c[r, j], 3, a, b)))
                for i in pick:
                    a = int(offs[t] + i)
                    got = sorted(zip(col[row_ptr[a]:row_ptr[a + 1]].tolist(), cost[row_ptr[a]:row_ptr[a + 1]].tolist()))
                    ok_arcs &= got == sorted(want[int(i)])
                    n_rows += 1; n_arcs += len(got)
        extra['arc_rows_vs_oracle'] = {'rows': n_rows, 'arcs': n_arcs, 'ok': bool(ok_arcs)}
        ok_assoc = bool(ok_arcs and extra.get('flow_certificate', {}).get('ok', False))
        what = 'optimality certificate over all arcs + sampled arc rows equal to the oracle (the oracle tracker does not finish at this size)'
    elif args.workload == 'c3' and world == 1:
        Pc = dict(orc.DEFAULTS)
        ref = orc.inference(frames_host, sd, P=Pc, yolo=list(yolo), assoc=args.assoc)
        frame_of = np.repeat(np.arange(len(cnt)), cnt)
        offs = np.concatenate([[0], np.cumsum(cnt)])
        got = {}
        for k, tid in enumerate(ad._track_flat):
            if tid >= 0:
                got.setdefault(int(tid), []).append((int(frame_of[k]), int(k - offs[frame_of[k]])))
        ok_assoc = [sorted(got[i]) for i in sorted(got)] == ref['trajs']
        if args.assoc == 'mcf':
            ok_assoc &= bool(ad.mcf_total_cost == ref['total_cost'])
        what = f'trajectories equal to the oracle ({args.assoc})'
    elif args.workload == 'c3':
        what = 'cross-rank hash of the trajectories (tracks_identical_on_all_ranks)'
    ok_cert = extra.get('flow_certificate', {}).get('ok', True)
    return {'verified': bool(ok_cnn and ok_det and ok_assoc is not False and ok_cert),
            'verify': dict({'cnn_frames': sample, 'cnn_max_rel_err': float(f'{err:.3e}'), 'cnn_ok': ok_cnn,
                            'detections_bit_exact_frames': len(ref_dets), 'detections_ok': ok_det,
                            'association': what, 'association_ok': ok_assoc}, **extra,
                           seconds=round(time.perf_counter() - t0, 1))}


def _profile_key(name):
    """'conv2 40>80 +pool' (bench table) / 'conv3x3_wino<40->80,s1,pool>' / 'conv<40->80,s1,pool>' (profiles) -> ('40>80', pool?)"""
    import re
    m = re.search(r'(\d+)-?>(\d+)', name)
    return (f'{m.group(1)}>{m.group(2)}', 'pool' in name) if m else None


def _read_profile_csv(path):
    """Rows of a committed profiles/*.csv as dicts; older pmc summaries wrote kernel names with unquoted commas."""
    import csv
    lines = open(path).read().splitlines()
    head = lines[0].split(',')
    rows = []
    for r in csv.reader(lines[1:]):
        if len(r) > len(head):
            r = [','.join(r[:len(r) - len(head) + 1])] + r[len(r) - len(head) + 1:]
        rows.append(dict(zip(head, r)))
    return rows


def _profile_set_matches(csv_path, args):
    """(ok, reason): a committed rocprofv3 summary may feed this line only if it measured THESE kernel sources and THIS workload:
    profiles/<tag>_meta.json (profiles/profile_meta.py, written by the collection scripts) holds the sha256 of the conv kernel
    sources and the bench.py arguments of the profiled command."""
    sys.path.insert(0, os.path.join(ROOT, 'profiles'))
    import profile_meta
    tag = os.path.basename(csv_path).rsplit('_', 1)[0]
    for cand in (f'{tag}_pmc_meta.json' if csv_path.endswith('_pmc.csv') else f'{tag}_meta.json', f'{tag}_meta.json'):
        path = os.path.join(ROOT, 'profiles', cand)
        if os.path.exists(path):
            break
    else:
        return False, f'{os.path.basename(csv_path)}: no {tag}_meta.json beside it (collected before round 4: sources and command unknown)'
    meta = json.load(open(path))
    if meta.get('sources_sha256') != profile_meta.sources_sha256(ROOT):
        return False, f'{cand}: the conv kernel sources have changed since this set was collected'
    mine = profile_meta.workload_key(['--workload', args.workload, '--assoc', args.assoc, '--arith', args.arith, '--input', args.input,
                                      '--size', str(args.size), '--frames', str(args.frames)])
    theirs = dict(meta.get('workload_key', {}))
    for k in ('size', 'frames'):                      # defaults spelled out
        if theirs.get(k) is None:
            theirs[k] = {'size': {'c4': '1024', 'c5': '1024'}.get(theirs.get('workload'), '512'),
                         'frames': {'c4': '132', 'c5': '68'}.get(theirs.get('workload'), '256')}[k]
    if theirs != mine:
        return False, f'{cand}: collected for {theirs}, this command is {mine}'
    return True, cand


def committed_counters(table, dom_name, winograd, wino_blocks, ms_per_step_cnn, args=None):
    """The north-star counters of this command from the committed rocprofv3 passes (profiles/, newest set that holds
    the dominant kernel): HBM bytes per launch and GB/s (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, FETCH doubled per
    the gfx950 correction) and the matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE / 8 XCDs x 1024
    SIMDs), for the dominant kernel and for the whole CNN of one pass. None where no committed profile matches."""
    import glob
    def is_wino(k):
        return winograd and k.split()[0] in wino_blocks
    def match(rows, bench_name):
        key, want_wino = _profile_key(bench_name), is_wino(bench_name)
        for r in rows:
            k = r['kernel']
            if 'fused' in bench_name or 'fused' in k:          # conv_s2_fused <-> 'conv0+1 5>20>40 s2 fused'
                if 'fused' in bench_name and 'fused' in k:
                    return r
                continue
            if 'conv' in k and _profile_key(k) == key and ('wino' in k) == want_wino:
                return r
        return None
    out = {'traffic': None, 'hbm_gbps': None, 'mfma_busy': None}
    kfiles = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_kernels.csv')))           # r01_* < r02* < r03*: the newest round last
    pfiles = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.csv')))
    why = []
    for f in reversed(kfiles):
        rows = _read_profile_csv(f)
        d = match(rows, dom_name)
        if d is None or 'FETCH_SIZE_KB_per_launch' not in d:
            continue
        ok, reason = _profile_set_matches(f, args) if args is not None else (True, '')
        if not ok:                                  # only the newest set that holds the kernel is a candidate: never fall back to an older one
            why.append(reason)
            break
        def nbytes(r):
            return (2 * float(r['FETCH_SIZE_KB_per_launch']) + float(r['WRITE_SIZE_KB_per_launch'])) * 1024
        out['traffic'] = int(nbytes(d))
        whole = 0.0
        for k in table:
            r = match(rows, k['name'])
            if r is not None and r['FETCH_SIZE_KB_per_launch'] not in ('nan', ''):
                whole += nbytes(r) * k['launches']
        out['hbm_gbps'] = {'kernel': round(nbytes(d) / (float(d['avg_us']) * 1e-6) / 1e9, 1),
                           'whole_cnn': round(whole / (ms_per_step_cnn * 1e-3) / 1e9, 1),
                           'peak': 8000.0, 'source': os.path.relpath(f, ROOT),
                           'how': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per launch (FETCH x 2, gfx950) / rocprofv3 average launch '
                                  'time (kernel); summed over the conv launches of a pass / the CNN time of this run (whole_cnn)'}
        out['traffic_detail'] = {'unit': 'HBM bytes per launch', 'source': os.path.relpath(f, ROOT),
                                 'fetch_kb_raw': float(d['FETCH_SIZE_KB_per_launch']), 'write_kb': float(d['WRITE_SIZE_KB_per_launch'])}
        break
    for f in reversed(pfiles):
        rows = _read_profile_csv(f)
        d = match(rows, dom_name)
        if d is None:
            continue
        ok, reason = _profile_set_matches(f, args) if args is not None else (True, '')
        if not ok:
            why.append(reason)
            break
        busy = lambda r: float(r['SQ_VALU_MFMA_BUSY_CYCLES']) / (float(r['GRBM_GUI_ACTIVE']) / 8 * 1024)
        num = den = 0.0
        for k in table:
            r = match(rows, k['name'])
            if r is not None:
                num += float(r['SQ_VALU_MFMA_BUSY_CYCLES']) * k['launches']
                den += float(r['GRBM_GUI_ACTIVE']) / 8 * 1024 * k['launches']
        out['mfma_busy'] = {'kernel': round(busy(d), 3), 'whole_cnn_convs': round(num / den, 3) if den else None,
                            'source': os.path.relpath(f, ROOT),
                            'how': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), rocprofv3 --pmc pass of this command '
                                   '(same kernel sources, same workload arguments: checked against the set\'s meta file)'}
        break
    if why:
        out['counters_withheld'] = why             # null counters with the reason, rather than stale ones under a fresh timing
    return out


def cpu_baseline_big(args, sd, synth):
    """Configs 4 / 5: the oracle on a bounded sample. c4: detection + global flow tracker on the first 24 detection frames.
    c5: detection on the first 8 detection frames, the masked path search of 16 sampled source detections (the oracle's search
    takes ~0.5 s per source and gap: a whole frame pair would take minutes), extrapolated to a frame's detections; the flow
    solve of the 8 frames on open-grid path lengths (its cost does not depend on where the lengths come from)."""
    from oracle import oracle as orc
    cores = min(len(os.sched_getaffinity(0)), 16)
    orc.set_threads(cores)
    H = W = args.size
    n = min(24 if args.workload == 'c4' else 8, args.frames - 4)
    frames = synth.synth_frames(args.frames, H, W, seed=0, t_range=(0, n + 4))
    Pc = dict(orc.DEFAULTS, MCF_MIN_FLOW=1)
    t = time.perf_counter()
    if args.workload == 'c4':
        orc.inference(frames, sd, P=Pc, assoc='mcf')
        dt = time.perf_counter() - t
        return {'value': round(n / dt, 3), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
                'sample': f'first {n} detection frames of the same synthetic timelapse: detection + path lengths + global flow tracker ({dt:.1f} s of CPU work), '
                          f'oracle/ (C + numpy restatement, OpenMP x{cores})'}
    mask = synth.corridor_mask(H, W, width=40, pitch=128)
    frames *= mask[None].astype(np.float32)
    dets, yolo = orc.detect_dataset(frames, sd, return_yolo=True)
    t_det = time.perf_counter() - t
    t = time.perf_counter()
    k = min(16, len(dets[0][0]))
    src = tuple(a[:k] for a in dets[0])
    for g in (1, 2):
        orc.path_matrix(src, dets[g], H, W, mask)
    t_path = (time.perf_counter() - t) / k
    t = time.perf_counter()
    orc.inference(frames, sd, P=Pc, yolo=yolo, assoc='mcf')                 # open-grid lengths: the solve's share only
    t_solve = time.perf_counter() - t
    per_frame = t_det / n + t_path * float(np.mean([len(d[0]) for d in dets])) + t_solve / n
    return {'value': round(1.0 / per_frame, 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'EXTRAPOLATED from a bounded sample: detection of the first {n} detection frames ({t_det:.1f} s), the masked path search of {k} source '
                      f'detections against both following frames ({t_path * k:.1f} s; x {np.mean([len(d[0]) for d in dets]):.0f} detections per frame), the flow solve of '
                      f'those frames ({t_solve:.1f} s); oracle/ (C + numpy restatement, OpenMP x{cores})'}


def cpu_baseline(args, sd, synth):
    """The CPU oracle on the first `cpu_frames` detection frames of the same timelapse."""
    from oracle import oracle as orc
    if args.big:
        return cpu_baseline_big(args, sd, synth)
    n = min(args.cpu_frames, args.frames - 4)
    cores = min(len(os.sched_getaffinity(0)), 16)          # a one-GPU box's CPU share
    orc.set_threads(cores)                                 # (the environment variable is read too early to matter here)
    frames = synth.synth_frames(args.frames, args.size, args.size, seed=0, t_range=(0, n + 4))
    Pc = dict(orc.DEFAULTS, MCF_MIN_FLOW=1)
    t = time.perf_counter()
    if args.workload == 'c3':
        orc.inference(frames, sd, P=Pc, assoc=args.assoc)
    else:
        orc.detect_dataset(frames, sd)
    dt = time.perf_counter() - t
    return {'value': round(n / dt, 3), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'{"all" if n == args.frames - 4 else "first"} {n} detection frames of the same synthetic timelapse ({dt:.1f} s of CPU work), '
                      f'oracle/ (C + numpy restatement, OpenMP x{cores})'}


if __name__ == '__main__':
    main()
