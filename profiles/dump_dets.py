#!/usr/bin/env python3
"""Dump the detections + arcs of the C3 bench timelapse (for offline experiments with the flow solver)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import axtrack_amd
from axtrack_amd import synth, params, hotpath as hp
from axtrack_amd.detections import transition_cost_table
frames = synth.synth_frames(256, 512, 512, seed=0)
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=252)
tl = axtrack_amd.Timelapse(frames, name='t')
P = params.load_parameters()
ad = axtrack_amd.AxonDetections(model, tl, P, None)
ad.detect_dataset()
cnt, conf, x, y = ad._host_dets()
np.savez_compressed('gpurun_out/c3_dets.npz', count=cnt, conf=conf[:, :160], x=x[:, :160], y=y[:, :160])
print('saved', cnt.sum())
