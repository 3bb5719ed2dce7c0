#!/usr/bin/env python3
"""Where the frame-to-frame kernel spends its time: builds a DIAGNOSTIC copy of the library (-DAXT_HUNG_STATS;
/tmp/libaxtrack_hstats.so), runs the headline workload once and prints, over the frame pairs of the gap-1 launch, the
s_memtime ticks (100 MHz) of the initialisation and of the searches, the searches and search steps per pair."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'axtrack_amd', 'csrc')
so = '/tmp/libaxtrack_hstats.so'
objs = [os.path.join(src, f) for f in os.listdir(src) if f.endswith('.o') and f != 'hungarian.o']
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-DAXT_HUNG_STATS',
                       '-c', os.path.join(src, 'hungarian.hip'), '-o', '/tmp/hungarian_stats.o'])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-o', so, '/tmp/hungarian_stats.o'] + objs)
os.environ['AXT_LIB_PATH'] = so
sys.path.insert(0, ROOT)
import torch
import axtrack_amd
from axtrack_amd import synth, params, _lib
frames = synth.synth_frames(256, 512, 512, seed=0)
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=252)
P = params.load_parameters()
P['ASSOCIATION'] = 'hungarian'
tl = axtrack_amd.Timelapse(frames, name='stats')
for _ in range(2):
    dets = axtrack_amd.inference(tl, model, None, P, None, None, None)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((4096, 8), np.uint64)
lib.axt_debug_hung_stats.argtypes = [ctypes.c_void_p]
assert lib.axt_debug_hung_stats(buf.ctypes.data) == 0
b = buf[buf[:, 4] > 0].astype(np.float64)
print(f'{len(b)} frame pairs; n {b[:, 4].mean():.1f} (max {b[:, 4].max():.0f}), m {b[:, 5].mean():.1f}')
for name, col in (('initialisation ticks', 0), ('search ticks', 1), ('searches', 2), ('search steps', 3)):
    print(f'{name:22s} mean {b[:, col].mean():9.1f}   max {b[:, col].max():9.1f}')
w = b[:, 1].argmax()
print(f'slowest pair: init {b[w, 0]:.0f} + searches {b[w, 1]:.0f} ticks (10 ns each), {b[w, 2]:.0f} searches, {b[w, 3]:.0f} steps, n {b[w, 4]:.0f}, m {b[w, 5]:.0f}')
print(f'ticks per step (all pairs): {b[:, 1].sum() / max(b[:, 3].sum(), 1):.2f}')
