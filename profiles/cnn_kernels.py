#!/usr/bin/env python3
"""Per-kernel HIP-event times of the detector alone (252 tile-forwards, 512x512x256). Used for A/B runs on one box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import axtrack_amd
from axtrack_amd import synth
frames = torch.from_numpy(synth.synth_frames(256, 512, 512, seed=0)).cuda()
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=252)
for _ in range(2):
    model.detect_frames(frames, [(0, 0)])
torch.cuda.synchronize()
model.set_profiling(True); model.read_profile()
R = 5
for _ in range(R):
    model.detect_frames(frames, [(0, 0)])
torch.cuda.synchronize()
prof = model.read_profile()
tot = sum(k['ms'] for k in prof) / R
print(' '.join(f"{k['name'].split()[0]}={k['ms'] / R:.3f}" for k in prof[:9]), f'total={tot:.3f}')
