#!/usr/bin/env python3
"""BASELINE configs 4 and 5 at FULL size on ONE GPU (the 8-GPU frame-sharded runs are the driver's; this exercises the same
path -- detection of every frame, global flow tracker over the whole timelapse -- unsharded, with stage times and the
size-independent properties the tests check at smaller sizes).
    python profiles/full_config.py c4     # synthetic 1024x1024x1024, all-ones mask
    python profiles/full_config.py c5     # synthetic 1024x1024x512 with the corridor mask (masked path costs)
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import axtrack_amd
from axtrack_amd import synth, params

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
T_all = int(sys.argv[2]) if len(sys.argv) > 2 else {'c4': 1024, 'c5': 512}[cfg]
out = {'config': cfg, 'frames_in': T_all, 'size': 1024}
def stamp(name, t0):
    torch.cuda.synchronize()
    out[name + '_s'] = round(time.perf_counter() - t0, 3)
    print(f'{name}: {out[name + "_s"]} s', flush=True)
    return time.perf_counter()
t = time.perf_counter()
frames = synth.synth_frames(T_all, 1024, 1024, seed=0)
mask = synth.corridor_mask(1024, 1024, width=40, pitch=128) if cfg == 'c5' else None
if mask is not None:
    frames *= mask[None].astype(np.float32)
t = stamp('synth', t)
P = params.load_parameters()
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=1024)
tl = axtrack_amd.Timelapse(frames, name=cfg, mask=mask)
del frames
ad = axtrack_amd.AxonDetections(model, tl, P, None)
ad.detect_dataset(); torch.cuda.synchronize()          # warm-up (tile list, allocations)
t = time.perf_counter()
ad.detect_dataset()
t = stamp('detect', t)
cnt, conf, x, y = ad._host_dets()
out.update(detection_frames=len(cnt), detections=int(cnt.sum()), max_per_frame=int(cnt.max()))
if mask is not None:
    ad._mask_dev()
    t = stamp('grid', t)
t = time.perf_counter()
ad.assign_ids()
t = stamp('assign_ids', t)
out.update(n_ids=ad.n_ids, total_cost=ad.mcf_total_cost)
out['frames_per_s_detect_plus_associate'] = round(len(cnt) / (out['detect_s'] + out['assign_ids_s']), 1)
# properties (tests/test_gpu_parity.py::test_full_size_properties): sorted, NMS distance, floor; node-disjoint tracks over
# increasing frames with gaps <= 2
ok = True
for f in range(0, len(cnt), 37):
    n = int(cnt[f])
    ok &= bool(np.all(np.diff(conf[f, :n].astype(np.float64)) <= 0)) and bool(conf[f, :n].min() >= np.float32(0.55))
    d2 = (x[f, :n, None] - x[f, None, :n]).astype(np.int64) ** 2 + (y[f, :n, None] - y[f, None, :n]).astype(np.int64) ** 2
    np.fill_diagonal(d2, 10 ** 9)
    ok &= bool(d2.min() >= 529)
track = ad._track_flat
offs = ad._offs
frame_of = np.searchsorted(offs, np.arange(len(track)), side='right') - 1
used = track >= 0
order = np.lexsort((frame_of[used], track[used]))
tr, fr = track[used][order], frame_of[used][order]
same = tr[1:] == tr[:-1]
gaps = (fr[1:] - fr[:-1])[same]
ok &= bool(np.all((gaps == 1) | (gaps == 2))) and P['MCF_MIN_FLOW'] <= ad.n_ids <= P['MCF_MAX_FLOW']
out['properties_ok'] = bool(ok)
out['used_detections'] = int(used.sum())
t = time.perf_counter()
df = ad.IDed_dets_all
t = stamp('ided_table', t)
out['ided_shape'] = list(df.shape)
print(json.dumps(out), flush=True)
