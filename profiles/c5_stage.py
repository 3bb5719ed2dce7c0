#!/usr/bin/env python3
"""Stage times of one GPU's share of BASELINE config 5 (1024x1024 frames, corridor mask, masked path costs, flow solve).
   python profiles/c5_stage.py [frames=16]        AXT_PATH_DEBUG=1 prints how many sources took which search"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import axtrack_amd
from axtrack_amd import synth, params

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def stamp(msg, t0):
    torch.cuda.synchronize()
    print(f'{msg}: {time.perf_counter() - t0:.3f} s', flush=True)
    return time.perf_counter()
t = time.perf_counter()
frames = synth.synth_frames(F + 4, 1024, 1024, seed=3)
mask = synth.corridor_mask(1024, 1024, width=40, pitch=128)
frames = frames * mask[None].astype(np.float32)
t = stamp('synth', t)
P = params.load_parameters()
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=64)
tl = axtrack_amd.Timelapse(frames, name='c5', mask=mask)
ad = axtrack_amd.AxonDetections(model, tl, P, None)
t = stamp('setup', t)
ad.detect_dataset()
t = stamp('detect', t)
cnt, conf, x, y = ad._host_dets()
on = sum(int(0 <= y[f, i] < 1024 and 0 <= x[f, i] < 1024 and mask[y[f, i], x[f, i]]) for f in range(len(cnt)) for i in range(cnt[f]))
print(f'detections {int(cnt.sum())}, on the mask {on}', flush=True)
grid = ad._mask_dev()
t = stamp('grid (components, off-cell fields)', t)
for rep in range(2):
    ad.assign_ids()
    t = stamp(f'assign_ids #{rep} (arcs + flow solve)', t)
print('tracks', ad.n_ids, flush=True)
# the arc build alone (masked path searches + CSR), as assign_ids calls it
from axtrack_amd import hotpath as hp
from axtrack_amd.detections import _cost_units_on_device
dmax, units = _cost_units_on_device(P, ad.max_px_assoc_dist, ad.device)
for rep in range(2):
    t = time.perf_counter()
    arcs = hp.build_arcs(ad.d_x, ad.d_y, ad.d_count, 1024, 1024, dmax, units, grid, ad.max_px_assoc_dist, ad.conn8)
    t = stamp(f'build_arcs #{rep} ({int(arcs[1].numel())} arcs)', t)
