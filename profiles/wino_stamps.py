#!/usr/bin/env python3
"""Where a chunk of the Winograd kernels spends its cycles: builds a DIAGNOSTIC copy of the library with s_memtime stamps
at the segment boundaries of wino_body's step() (-DAXT_WINO_STAMPS; /tmp/libaxtrack_stamps.so), runs the detector on
config 3 and prints, per kind of wave, the average cycles per chunk of every segment of ONE Winograd instantiation.
Shares, not lengths: the stamps' fences forbid overlaps the real kernel has.
    python profiles/wino_stamps.py [CIN [POOL]]     default 40 1 = conv block 2 (40 -> 80 + pool); 80 0 = blocks 4 / 7"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'axtrack_amd', 'csrc')
so = '/tmp/libaxtrack_stamps.so'
objs = [os.path.join(src, f) for f in os.listdir(src) if f.endswith('.o') and f != 'cnn.o']
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-DAXT_WINO_STAMPS' ,
                       f'-DAXT_WINO_STAMP_CIN={sys.argv[1] if len(sys.argv) > 1 else 40}', f'-DAXT_WINO_STAMP_POOL={sys.argv[2] if len(sys.argv) > 2 else 1}',
                       '-c', os.path.join(src, 'cnn.hip'), '-o', '/tmp/cnn_stamps.o'])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-o', so, '/tmp/cnn_stamps.o'] + objs)
os.environ['AXT_LIB_PATH'] = so
sys.path.insert(0, ROOT)
import torch
import axtrack_amd
from axtrack_amd import synth, _lib
frames = torch.from_numpy(synth.synth_frames(256, 512, 512, seed=0)).cuda()
model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=252)
for _ in range(3):
    model.detect_frames(frames, [(0, 0)])
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((1024, 8, 8), np.uint64)
lib.axt_debug_wino_stamps.argtypes = [ctypes.c_void_p]
assert lib.axt_debug_wino_stamps(buf.ctypes.data) == 0
used = buf[:, 0, 5] > 0
b = buf[used].astype(np.float64)
steps = b[:, :, 5:6]
names = ['transform + V store', 'MFMAs (+ load issue)', 'wait for DMA', 'epilogue (per chunk)', 'barrier', ]
print(f'{used.sum()} workgroups, {steps.mean():.1f} chunks each; cycles per chunk (s_memtime ticks)')
for kind, sl in (('transform waves 0-3 (2 channel blocks)', slice(0, 4)), ('waves 4-7 (3 channel blocks)', slice(4, 8))):
    per = (b[:, sl, :5] / steps[:, sl]).mean(axis=(0, 1))
    wait = (b[:, sl, 6] / steps[:, sl, 0]).mean()
    print(kind, f'wait for the patches: {wait:.0f} | ' + ' | '.join(f'{n}: {v:.0f}' for n, v in zip(names, per)), f' | sum {per.sum() + wait:.0f}')
