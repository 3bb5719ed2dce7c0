#!/usr/bin/env python3
"""Fine-grained wall-clock breakdown of one hot-path pass (synchronised between stages), to see where
the time outside the big kernels goes. Run on the GPU box: python profiles/stage_timing.py [mcf|hungarian]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import axtrack_amd
from axtrack_amd import synth, params, hotpath as hp
from axtrack_amd.detections import transition_cost_table

mode = sys.argv[1] if len(sys.argv) > 1 else 'hungarian'
frames = synth.synth_frames(256, 512, 512, seed=0)
sd = synth.synth_state_dict(42)
P = params.load_parameters(); P['ASSOCIATION'] = mode
model = axtrack_amd.Detector(sd, max_batch=252)
tl = axtrack_amd.Timelapse(frames, name='t')
def T(name, fn, acc):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    acc.setdefault(name, []).append((time.perf_counter() - t) * 1e3); return r
acc = {}
for it in range(6):
    ad = axtrack_amd.AxonDetections(model, tl, P, None)
    tiles = T('tile_occupancy', lambda: hp.tile_occupancy(tl.frames), acc)
    yolo = T('cnn_forward', lambda: model.detect_frames(tl.frames, tiles, 0, tl.sizet), acc)
    d = T('decode_nms', lambda: hp.decode_stitch_nms(yolo, tiles, float(np.float32(0.55)), 23), acc)
    ad.tile_yx, ad._yolo = tiles, yolo
    ad.d_conf, ad.d_x, ad.d_y, ad.d_count = d
    ad._host = None; ad._det_tables = None
    T('host_dets (D2H)', ad._host_dets, acc)
    T('assign (solve)', ad._assign_IDs_to_detections, acc)
    ad._solved = True; ad._ided_tables = None
    T('IDed_dets_all', ad._agg_all_IDed_dets, acc)
    T('whole inference()', lambda: axtrack_amd.inference(tl, model, None, P, None, None, None), acc)
for k, v in acc.items():
    print(f'{k:22s} {np.median(v[1:]):8.3f} ms')
