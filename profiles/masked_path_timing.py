#!/usr/bin/env python3
"""Time axt_path_cost on a masked 1024x1024 grid (BASELINE config 5 shape): one frame pair, ~300 x 300 detections."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from axtrack_amd import synth, hotpath as hp
H = W = 1024
mask = synth.corridor_mask(H, W, 40, 128)
rng = np.random.default_rng(0)
n = 300
ys, xs = np.nonzero(mask)
pick = rng.choice(len(ys), 2 * n, replace=False)
xa, ya = xs[pick[:n]], ys[pick[:n]]
xb = np.clip(xa + rng.integers(-6, 7, n), 0, W - 1); yb = np.clip(ya + rng.integers(-6, 7, n), 0, H - 1)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.int32)).cuda()
m = torch.as_tensor(mask.astype(np.uint8)).cuda()
args = (dev(xa), dev(ya), dev(xb), dev(yb), H, W, m, 500, False)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    D = hp.path_cost(*args)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f'path_cost masked {n}x{n} @1024^2: {dt * 1e3:.1f} ms  (reached {(D < 500).float().mean().item():.2f})')
