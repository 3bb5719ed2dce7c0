import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import axtrack_amd
from axtrack_amd import synth, _lib, hotpath as hp
from oracle import oracle as orc
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (530, 701)
NF = int(os.environ.get('NF', 6))
frames = synth.synth_frames(NF, H, W, seed=17)
sd = synth.synth_state_dict(42)
fr = torch.from_numpy(frames).cuda()
keep = hp.tile_occupancy(fr)
model = axtrack_amd.Detector(sd, max_batch=16)
model.detect_frames(fr, keep)
torch.cuda.synchronize()
lib = _lib.load()
n_items = (NF - 4) * len(keep)
out = np.zeros(n_items * 20 * 256 * 256, np.float32)
lib.axt_debug_act.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
assert lib.axt_debug_act(model._h, 0, out.ctypes.data, out.size) == 0
out = out.reshape(n_items, 20, 256, 256)
L = orc.lib()
pre = 'ConvNet.ConvBlock_0.'
arrs = [np.ascontiguousarray(sd[pre + k], np.float32) for k in ('conv.weight', 'conv.bias', 'batchnorm.weight', 'batchnorm.bias', 'batchnorm.running_mean', 'batchnorm.running_var')]
for t in range(NF - 4):
    X = np.ascontiguousarray(orc.frame_tile_stack(frames, t, keep), np.float32)      # [n_tiles,5,512,512]
    ref = np.empty((len(keep), 20, 256, 256), np.float32)
    L.orc_conv3x3_bn_lrelu(orc._p(X), len(keep), 5, 512, 512, *[orc._p(a) for a in arrs], 20, 2, ctypes.c_float(0.1), orc._p(ref))
    got = out[t * len(keep):(t + 1) * len(keep)]
    d = np.abs(got - ref)
    bad = np.argwhere(d > 1e-4 + 1e-4 * np.abs(ref))
    print(f't={t}: bad {len(bad)} of {ref.size}, max abs {d.max():.4g}')
    if len(bad):
        for k in range(len(keep)):
            b = bad[bad[:, 0] == k]
            if len(b):
                print(f'  tile {keep[k]}: {len(b)} bad; ch {np.unique(b[:,1])}; per 16-row band {np.bincount(b[:,2] // 16, minlength=16)}; col%16 {np.unique(b[:,3] % 16)}; row%4 {np.unique(b[:,2] % 4)}')
