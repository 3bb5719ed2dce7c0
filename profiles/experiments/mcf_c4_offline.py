"""The flow solve of BASELINE config 4 without a GPU: the detections of the synthetic 1024x1024x1024 timelapse (c4_dets.npz, dumped on
the GPU box by profiles/dump_full_dets.py c4) -> the network as assign_ids builds it -> axt_mcf_solve.
    python profiles/experiments/mcf_c4_offline.py [frames]      AXT_MCF_DEBUG=1 prints the tree; REPEAT=n"""
import os, sys, time, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from axtrack_amd import hotpath as hp
from tests import helpers
d = np.load(os.path.join(HERE, 'c4_dets.npz'))
F = int(sys.argv[1]) if len(sys.argv) > 1 else len(d['count'])
t=time.perf_counter()
if os.path.exists(f'/tmp/c4_net_{F}.npz'):
    z = np.load(f'/tmp/c4_net_{F}.npz'); net = [z[k] for k in ('obs','en','ex','row_ptr','col','cost')]
else:
    net = helpers.open_grid_network(d['count'][:F], d['x'][:F].astype(np.int64), d['y'][:F].astype(np.int64), d['conf'][:F], 1024, 1024)[:6]
    np.savez(f'/tmp/c4_net_{F}.npz', obs=net[0], en=net[1], ex=net[2], row_ptr=net[3], col=net[4], cost=net[5])
print('network %.1f s: %d dets %d arcs' % (time.perf_counter()-t, len(net[0]), len(net[4])), flush=True)
for _ in range(int(os.environ.get('REPEAT', 1))):
    t=time.perf_counter(); r = hp.mcf_solve(*net, 5, 100000); print('solve %.1f ms tracks %d cost %d' % (1e3*(time.perf_counter()-t), r[2], r[3]), flush=True)
