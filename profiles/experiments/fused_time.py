#!/usr/bin/env python3
"""HIP-event time of the fused front kernel (the default) for the library in AXT_LIB_PATH: 252 tile-forwards."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import axtrack_amd
from axtrack_amd import synth
frames = torch.from_numpy(synth.synth_frames(256, 512, 512, seed=0)).cuda()
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=252)
for _ in range(2):
    model.detect_frames(frames, [(0, 0)])
torch.cuda.synchronize()
model.set_profiling(True); model.read_profile()
for _ in range(5):
    model.detect_frames(frames, [(0, 0)])
torch.cuda.synchronize()
prof = model.read_profile()
print(os.path.basename(os.environ.get('AXT_LIB_PATH', 'tree')), f"front={prof[0]['ms'] / 5:.3f} ms", flush=True)
