#!/usr/bin/env python3
"""Host-side profile of one config-3 pass (cProfile over 30 passes): where the Python / ctypes time between kernel launches
goes.    python profiles/host_profile.py      (on the GPU box)"""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import axtrack_amd
from axtrack_amd import synth, params

frames = synth.synth_frames(256, 512, 512, seed=0)
sd = synth.synth_state_dict(42)
model = axtrack_amd.Detector(sd, max_batch=252)
tl = axtrack_amd.Timelapse(frames, name='c3')
P = dict(params.load_parameters(), ASSOCIATION=(sys.argv[1] if len(sys.argv) > 1 else 'hungarian'))

def step():
    ad = axtrack_amd.AxonDetections(model, tl, P, None)
    ad.detect_dataset(cache=None)
    ad.assign_ids(None, None)
    return ad

for _ in range(5):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
print('ms per pass %.3f' % ((time.perf_counter() - t) / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
