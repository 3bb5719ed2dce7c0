#!/usr/bin/env python3
"""One-off randomized differential run of the whole path against the oracle (more shapes / seeds / modes than the
committed tests): python profiles/fuzz_parity.py [n_cases] [rng_seed]. Everything after the CNN must be identical when the oracle
is fed the HIP YOLO grids; the CNN itself is compared with the stated tolerance."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import axtrack_amd
from axtrack_amd import synth, params
from oracle import oracle as orc
from helpers import tracks_from_next

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
sd = synth.synth_state_dict(42)
model = axtrack_amd.Detector(sd, max_batch=40)
bad = 0
for case in range(n_cases):
    H, W = [(512, 512), (530, 701), (1024, 512), (700, 1100), (512, 1028), (1024, 1024)][case % 6]
    T = int(rng.integers(7, 11))
    seed = int(rng.integers(0, 10 ** 6))
    mode = ['mcf', 'hungarian', 'mcf+mask', 'mcf+vis'][case % 4]
    frames = synth.synth_frames(T, H, W, seed=seed)
    if 'vis' in mode:
        frames = frames * np.float32(0.5)
    mask = synth.corridor_mask(H, W, 40, 128) if 'mask' in mode else None
    P = params.load_parameters(); P['ASSOCIATION'] = mode.split('+')[0]; P['MCF_MIN_FLOW'] = 1
    Po = dict(orc.DEFAULTS, MCF_MIN_FLOW=1)
    if 'vis' in mode:
        P['MCF_VIS_SIM_WEIGHT'] = Po['MCF_VIS_SIM_WEIGHT'] = 0.3
    if mask is not None:
        frames = frames * mask[None].astype(np.float32)
    t0 = time.time()
    tl = axtrack_amd.Timelapse(frames, name='synth', mask=mask)
    ad = axtrack_amd.inference(tl, model, None, P, None, None, None)
    yolo = ad._yolo.cpu().numpy()
    ref = orc.inference(frames, sd, mask=mask, P=Po, yolo=list(yolo), assoc=P['ASSOCIATION'])
    cnt, conf, x, y = ad._host_dets()
    ok = all(int(cnt[t]) == len(rc) and np.array_equal(conf[t, :len(rc)], rc) and np.array_equal(x[t, :len(rc)], rx)
             and np.array_equal(y[t, :len(rc)], ry) for t, (rc, rx, ry) in enumerate(ref['dets']))
    got = tracks_from_next(np.zeros(len(ad._track_flat)), ad._track_flat, ad._offs) if ad._solved else None
    ok_tr = got == ref['trajs']
    # CNN against the oracle's own forward pass on two frames
    keep = orc.kept_tiles(frames)
    cerr = max(float(np.abs(yolo[t] - orc.cnn_forward(sd, orc.frame_tile_stack(frames, t, keep))).max()) for t in (0, T - 5))
    print(f'case {case}: {H}x{W}x{T} {mode:10s} seed {seed:6d} dets {int(cnt.sum()):5d} ids {ad.n_ids}  detections {"ok" if ok else "DIFF"}  '
          f'tracks {"ok" if ok_tr else "DIFF"}  cnn max err {cerr:.2e}  ({time.time() - t0:.1f} s)', flush=True)
    bad += (not ok) + (not ok_tr) + (cerr > 2e-4)
print('FAILED' if bad else 'all cases agree')
sys.exit(1 if bad else 0)
