#!/usr/bin/env python3
"""Host-side profile of config-3 passes over FRESH Timelapse objects (cProfile over 20 passes): what a timelapse that is processed
once pays beyond the steady-state pass of bench.py.    python profiles/fresh_profile.py      (on the GPU box)"""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import axtrack_amd
from axtrack_amd import synth, params, detections

frames = synth.synth_frames(256, 512, 512, seed=0)
sd = synth.synth_state_dict(42)
model = axtrack_amd.Detector(sd, max_batch=252)
tl0 = axtrack_amd.Timelapse(frames, name='c3')
P = dict(params.load_parameters(), ASSOCIATION='hungarian')

def step(fresh):
    tl = axtrack_amd.Timelapse(tl0.frames, name='c3') if fresh else tl0
    if fresh: detections._IDS_GUESS.clear()
    ad = axtrack_amd.AxonDetections(model, tl, P, None)
    ad.detect_dataset(cache=None)
    ad.assign_ids(None, None)
    return ad

for fresh in (False, True):
    for _ in range(5): step(fresh)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): step(fresh)
    torch.cuda.synchronize()
    print('%s: ms per pass %.3f' % ('fresh' if fresh else 'steady', (time.perf_counter() - t) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20): step(True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
