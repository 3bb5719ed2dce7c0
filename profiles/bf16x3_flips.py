#!/usr/bin/env python3
"""Characterises one CNN_ARITH arithmetic against another on BASELINE config 3 (512x512x256, 252 detection frames):
max |yolo - oracle| on sampled frames for both, max |B - A| over all grids, and how many detections differ after
decode / 0.55 cut / NMS (anchors moved by a rounding tie, confidences crossing the floor).
    python profiles/bf16x3_flips.py [A B]      (on the GPU box; default: f32_direct bf16x3; the keys keep the names
                                                 f32 / bf16x3 of the first study: f32 = A, bf16x3 = B)"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import axtrack_amd
from axtrack_amd import synth, params
from oracle import oracle as orc

frames = synth.synth_frames(256, 512, 512, seed=0)
sd = synth.synth_state_dict(42)
model = axtrack_amd.Detector(sd, max_batch=252)
tl = axtrack_amd.Timelapse(frames, name='c3')
res = {}
A, Bn = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ('f32_direct', 'bf16x3')
for arith, name in (('f32', A), ('bf16x3', Bn)):
    P = dict(params.load_parameters(), ASSOCIATION='hungarian', CNN_ARITH=name)
    ad = axtrack_amd.inference(tl, model, None, P, None, None, None)
    res[arith] = (ad._yolo.cpu().numpy(), ad._host_dets(), ad.n_ids)
sample = [0, 1, 63, 126, 127, 128, 200, 251]
ref = np.stack([orc.cnn_forward(sd, orc.frame_tile_stack(frames, t, [(0, 0)])) for t in sample])
out = {'frames': 252, 'oracle_sample': sample, 'A (keys f32)': A, 'B (keys bf16x3)': Bn}
for arith in res:
    out[f'max_abs_err_vs_oracle_{arith}'] = float(np.abs(res[arith][0][sample] - ref).max())
y32, yb = res['f32'][0], res['bf16x3'][0]
out['max_abs_diff_bf16x3_vs_f32'] = float(np.abs(yb - y32).max())
out['mean_abs_diff_bf16x3_vs_f32'] = float(np.abs(yb - y32).mean())
c32, cb = res['f32'][1], res['bf16x3'][1]
moved = crossed = same = 0
for t in range(252):
    a = {(int(x), int(y)) for x, y in zip(c32[2][t, :c32[0][t]], c32[3][t, :c32[0][t]])}
    b = {(int(x), int(y)) for x, y in zip(cb[2][t, :cb[0][t]], cb[3][t, :cb[0][t]])}
    same += len(a & b)
    only_a, only_b = a - b, b - a
    near = sum(1 for (x, y) in only_a if any(abs(x - u) <= 1 and abs(y - v) <= 1 for (u, v) in only_b))
    moved += near
    crossed += (len(only_a) - near) + (len(only_b) - near)
out.update(detections_f32=int(c32[0].sum()), detections_bf16x3=int(cb[0].sum()), identical_detections=same,
           anchors_moved_by_one_pixel=moved, detections_gained_or_lost=crossed,
           conf_floor_crossings=int(((y32[..., 0] >= np.float32(0.55)) != (yb[..., 0] >= np.float32(0.55))).sum()),
           n_ids_f32=res['f32'][2], n_ids_bf16x3=res['bf16x3'][2])
print(json.dumps(out, indent=1))
