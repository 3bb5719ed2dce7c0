#!/usr/bin/env python3
"""Times axt_mcf_solve (host C++) without a GPU, on
  static  the flow network of the C3 bench timelapse (19 340 detections, 553 k arcs; tests/data/c3_dets.npz), optionally
          tiled `rep` times in time, or
  moving  a scene of moving growth cones with births, deaths, misses and clutter (synth.synth_detections) at the size
          of BASELINE config 3 (c3) or 4 (c4).
    python profiles/mcf_timing.py [static [rep] | moving c3|c4 [seed]]        AXT_MCF_DEBUG=1 prints search statistics"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from axtrack_amd import hotpath as hp
from tests import helpers


def static_network(rep=1):
    d = np.load(os.path.join(ROOT, 'tests', 'data', 'c3_dets.npz'))
    t = np.arange(len(d['count']) * rep) % len(d['count'])
    return helpers.open_grid_network(d['count'][t], d['x'][t], d['y'][t], d['conf'][t], 512, 512)[:6]


MOVING = {'c3': (252, 512, 90), 'c4': (1020, 1024, 380)}

if __name__ == '__main__':
    kind = sys.argv[1] if len(sys.argv) > 1 else 'static'
    if kind == 'moving':
        F, size, alive = MOVING[sys.argv[2] if len(sys.argv) > 2 else 'c3']
        net = helpers.moving_network(F, size, alive, seed=int(sys.argv[3]) if len(sys.argv) > 3 else 0)[:6]
        max_flow = 100000
    else:
        rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1
        net = static_network(rep)
        max_flow = 450 * rep
    print(f'{len(net[0])} detections, {len(net[4])} arcs')
    times = []
    for _ in range(int(os.environ.get('REPEAT', 5))):
        t = time.perf_counter()
        res = hp.mcf_solve(*net, 5, max_flow)
        times.append(1e3 * (time.perf_counter() - t))
    print(f'solve min {min(times):.1f} ms, median {sorted(times)[len(times) // 2]:.1f} ms of {len(times)} runs; tracks {res[2]}  cost {res[3]}')
