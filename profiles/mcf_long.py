#!/usr/bin/env python3
"""Global flow solve on ONE long synthetic timelapse (default 1024 frames of 512x512), at several thread counts of the
time-blocked solver (AXT_MCF_THREADS). Needs the GPU (detection).   python profiles/mcf_long.py [frames] [size]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import axtrack_amd
from axtrack_amd import synth, params

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
t0 = time.perf_counter()
frames = synth.synth_frames(T + 4, S, S, seed=0)
print(f'synthesised {T + 4} frames in {time.perf_counter() - t0:.1f} s', flush=True)
P = params.load_parameters()
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=256)
tl = axtrack_amd.Timelapse(frames, name='long')
ad = axtrack_amd.AxonDetections(model, tl, P, None)
ad.detect_dataset()
torch.cuda.synchronize()
print('detections', int(ad.d_count.sum()), flush=True)
ref = None
for th in (1, 2, 4, 8, 16):
    os.environ['AXT_MCF_THREADS'] = str(th)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        ad.assign_ids()
        best = min(best, time.perf_counter() - t0)
    if ref is None:
        ref = (ad._track_flat.copy(), ad.mcf_total_cost)
    same = np.array_equal(ref[0], ad._track_flat) and ref[1] == ad.mcf_total_cost
    print(f'threads {th:2d}: assign_ids {1e3 * best:.1f} ms, tracks {ad.n_ids}, identical to 1 thread: {same}', flush=True)
