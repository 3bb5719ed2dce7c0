import os, sys, ctypes, subprocess
import numpy as np
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    import torch, axtrack_amd
    from axtrack_amd import synth, _lib
    frames = torch.from_numpy(synth.synth_frames(12, 512, 512, seed=3)).cuda()
    model = axtrack_amd.Detector(synth.synth_state_dict(42), max_batch=16)
    model.detect_frames(frames, [(0, 0)])
    torch.cuda.synchronize()
    lib = _lib.load()
    out = np.zeros(8 * 20 * 256 * 256, np.float32)
    lib.axt_debug_act.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    print(lib.axt_debug_act(model._h, 0, out.ctypes.data, out.size))
    np.save(sys.argv[1], out.reshape(160, 256, 256))
else:
    for tag, d in (('0', '0'), ('8', '8'), ('0b', '0'), ('8b', '8')):
        subprocess.run([sys.executable, __file__, f'/tmp/c0_{tag}.npy'], env=dict(os.environ, AXT_DBG=d), check=True)
    for x, y in (('0', '0b'), ('8', '8b'), ('0', '8b')):
        print(x, y, 'differing:', int((np.load(f'/tmp/c0_{x}.npy') != np.load(f'/tmp/c0_{y}.npy')).sum()))
    a, b = np.load('/tmp/c0_0.npy'), np.load('/tmp/c0_8.npy')
    bad = np.argwhere(a != b)
    print('differing:', len(bad), 'of', a.size)
    if len(bad):
        print('channels', np.unique(bad[:, 0]))
        print('rows', np.unique(bad[:, 1])[:40], '...')
        print('cols', np.unique(bad[:, 2])[:80])
        print('max abs', np.abs(a - b).max())
