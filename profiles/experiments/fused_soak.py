#!/usr/bin/env python3
"""Soak of the fused front kernel: the headline workload's CNN N times (default 300), every pass's YOLO grids compared bit
for bit with the first pass's and, once, with the separate kernels' within 2e-5 -- an LDS-DMA ordering bug would show as a
rare difference."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import axtrack_amd
from axtrack_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for (T, H, W, tiles) in [(256, 512, 512, [(0, 0)]), (40, 1100, 1032, [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2)])]:
    frames = torch.from_numpy(synth.synth_frames(T, H, W, seed=1)).cuda()
    model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=(T - 4) * len(tiles))
    model.set_fused_front(False)
    sep = model.detect_frames(frames, tiles)
    model.set_fused_front(True)
    first = model.detect_frames(frames, tiles)
    assert float((first - sep).abs().max()) < 2e-5
    diff = torch.zeros((), dtype=torch.int64, device='cuda')
    for _ in range(N):
        y = model.detect_frames(frames, tiles)
        diff += (y.view(torch.int32) != first.view(torch.int32)).sum()
    torch.cuda.synchronize()
    print(f'{T}x{H}x{W}, {len(tiles)} tiles per frame: {N} passes, {int(diff)} differing values', flush=True)
    bad += int(diff)
sys.exit(1 if bad else 0)
