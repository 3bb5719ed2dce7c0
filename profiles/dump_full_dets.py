#!/usr/bin/env python3
"""Dump the detections of BASELINE config 4 / 5 at full size (for offline work on the flow solver: the host solver
runs without a GPU, the detections need the CNN).
    python profiles/dump_full_dets.py c4|c5 [frames_in]   ->  gpurun_out/<cfg>_dets.npz  (count, conf, x, y; masked: arcs too)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import axtrack_amd
from axtrack_amd import synth, params

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
T_all = int(sys.argv[2]) if len(sys.argv) > 2 else {'c4': 1024, 'c5': 512}[cfg]
frames = synth.synth_frames(T_all, 1024, 1024, seed=0)
mask = synth.corridor_mask(1024, 1024, width=40, pitch=128) if cfg == 'c5' else None
if mask is not None:
    frames *= mask[None].astype(np.float32)
P = params.load_parameters()
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=1024)
tl = axtrack_amd.Timelapse(frames, name=cfg, mask=mask)
del frames
ad = axtrack_amd.AxonDetections(model, tl, P, None)
ad.detect_dataset()
cnt, conf, x, y = ad._host_dets()
m = int(cnt.max())
out = dict(count=cnt, conf=conf[:, :m], x=x[:, :m].astype(np.int16), y=y[:, :m].astype(np.int16))
if mask is not None:                      # the masked path lengths cannot be rebuilt on the CPU in reasonable time: keep the network
    ad.assign_ids()
    net = ad._last_network if hasattr(ad, '_last_network') else None
    if net is not None:
        out.update(row_ptr=net['row_ptr'], col=net['col'], cost=net['cost'], obs=net['obs'], entry=net['entry'], exit=net['exit'])
os.makedirs('gpurun_out', exist_ok=True)
np.savez_compressed(f'gpurun_out/{cfg}_dets.npz', **out)
print('saved', cfg, int(cnt.sum()), 'detections, max per frame', m)
