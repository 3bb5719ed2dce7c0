#!/usr/bin/env python3
"""Time the masked arc builder at BASELINE config 5 shape (1024x1024 corridor mask, ~300 detections per frame)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from axtrack_amd import synth, params, hotpath as hp
from axtrack_amd.detections import transition_cost_table
H = W = 1024
mask = synth.corridor_mask(H, W, 40, 128)
rng = np.random.default_rng(0)
F, n, cap = 32, 300, 576
ys, xs = np.nonzero(mask)
k = rng.choice(len(ys), n, replace=False)
x = np.zeros((F, cap), np.int32); y = np.zeros((F, cap), np.int32)
px, py = xs[k].astype(np.int64), ys[k].astype(np.int64)
for t in range(F):
    qx = np.clip(px + rng.integers(-3, 4, n), 0, W - 1); qy = np.clip(py + rng.integers(-3, 4, n), 0, H - 1)
    ok = mask[qy, qx]
    px, py = np.where(ok, qx, px), np.where(ok, qy, py)          # stay on the mask
    x[t, :n], y[t, :n] = px, py
cnt = np.full(F, n, np.int32)
table, dmax = transition_cost_table(params.DEPLOYED)
units = np.where(np.isfinite(table), np.rint(table * 1e6), 0).astype(np.int64)
grid = hp.Grid(mask)
d = lambda a: torch.as_tensor(a).cuda()
X, Y, C = d(x), d(y), d(cnt)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    row_ptr, col, length, gap, cost = hp.build_arcs(X, Y, C, H, W, dmax, units, grid)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'masked build_arcs: {F} frames x {n} dets @1024^2: {dt * 1e3:.1f} ms ({dt / F * 1e3:.2f} ms/frame), {col.numel()} arcs')
torch.cuda.synchronize(); t0 = time.perf_counter()
hp.build_arcs(X, Y, C, H, W, dmax, units, None)
torch.cuda.synchronize(); print(f'open-grid build_arcs for comparison: {(time.perf_counter() - t0) * 1e3:.1f} ms')
