#!/usr/bin/env python3
"""conv_s2_fused (Detector.set_fused_front) against the two separate stride-2 kernels: bit-equality of the YOLO grids on frames whose
edges cut the tiles, then per-kernel HIP-event times on the headline workload."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import axtrack_amd
from axtrack_amd import synth

def make(fused, mb):
    d = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=mb)
    d.set_fused_front(fused)
    return d

ok = True
for (T, H, W, tiles) in [(9, 512, 512, [(0, 0)]), (12, 700, 904, [(0, 0), (0, 1), (1, 0), (1, 1)]), (7, 300, 260, [(0, 0)]),
                         (8, 1100, 1032, [(0, 0), (1, 1), (2, 2), (0, 2), (2, 0)])]:
    frames = torch.from_numpy(synth.synth_frames(T, H, W, seed=3)).cuda()
    a = make(False, 64).detect_frames(frames, tiles).cpu().numpy()
    b = make(True, 64).detect_frames(frames, tiles).cpu().numpy()
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32)) or bool(np.abs(a - b).max() < 2e-5)
    print(f'{T}x{H}x{W} tiles={len(tiles)}: equal={same} bits={np.array_equal(a.view(np.uint32), b.view(np.uint32))} maxdiff={np.abs(a - b).max():.3g} finite={np.isfinite(b).all()}', flush=True)
    ok &= same
frames = torch.from_numpy(synth.synth_frames(256, 512, 512, seed=0)).cuda()
for fused in (False, True, False, True):
    model = make(fused, 252)
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
    print(f'fused={int(fused)}', ' '.join(f"{k['name'].split()[0]}={k['ms'] / R:.3f}" for k in prof[:9]), f'total={tot:.3f}', flush=True)
sys.exit(0 if ok else 1)
