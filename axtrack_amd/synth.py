"""Deterministic synthetic inputs for the AxTrack hot path (no reference code involved).

The reference ships neither weights nor example data, so fixtures, tests
and bench.py all use seeded synthetic timelapses and detector weights. Everything here is
derived from the raw PCG64 bit stream (guaranteed stable across numpy versions) with
explicit arithmetic, so the same seed gives the same bytes in this container and on the
GPU box.

Shapes follow the reference's contracts:
  * frames: what `Timelapse.X[:, 0]` holds after preprocessing (Timelapse.py:426-433),
    dense f32 [T_all, H, W], ~98 % zeros (Timelapse.py:271-274).
  * state_dict: the key set of YOLO_AXTrack for the deployed ARCHITECTURE
    (model.py:85-117; deployed_model/params.txt:34).
"""
import numpy as np

# deployed ARCHITECTURE (deployed_model/params.txt:34): (kernel, cout, stride, groups) or 'M'
CONV_ARCH = [(3, 20, 2, 1), (3, 40, 2, 1), (3, 80, 1, 1), 'M', (3, 80, 1, 1), (3, 80, 1, 1), 'M',
             (3, 80, 1, 1), (3, 80, 1, 1), 'M', (3, 160, 1, 1)]
FC_ARCH = [1024, 1024]
IN_CHANNELS = 5            # 1 * (2*TEMPORAL_CONTEXT + 1), core_functionality.py:66-67
TILESIZE = 512
SX = SY = 12


def _raw(seed, n):
    return np.random.PCG64(int(seed)).random_raw(int(n))


def uniform01(seed, shape):
    """U[0,1) float64 from the top 53 bits of the raw PCG64 stream."""
    n = int(np.prod(shape))
    u = (_raw(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def normal(seed, shape):
    """N(0,1) float64 by Box-Muller on two independent raw streams."""
    n = int(np.prod(shape))
    u1 = uniform01(seed * 2 + 1, n)
    u2 = uniform01(seed * 2 + 2, n)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return z.reshape(shape)


def synth_frames(T_all, H, W, seed=0, n_blobs=None, bg_frac=0.02, t_range=None):
    """Preprocessed-looking timelapse, f32 [T_all, H, W].
    t_range=(a, b) renders only frames a..b-1 of that timelapse (frame-sharded ranks), identical
    to slicing the full result.

    ~bg_frac of the pixels carry background speckle in (0, 3]; `n_blobs` Gaussian blobs
    (sigma 4 px, peak ~8) random-walk <= 10 px/frame as growth cones. Every 512x512 tile is
    non-empty at every t (the reference drops tiles that are empty at all t,
    Timelapse.py:551,558).
    """
    if n_blobs is None:
        n_blobs = max(8, (H * W) // 5243)          # ~50 per 512x512
    ta, tb = (0, T_all) if t_range is None else t_range
    frames = np.zeros((tb - ta, H, W), np.float32)
    # background speckle, different every frame
    for t in range(ta, tb):
        u = uniform01(seed * 1000003 + 17 * t + 1, (H, W))
        v = uniform01(seed * 1000003 + 17 * t + 2, (H, W))
        frames[t - ta] = np.where(u < bg_frac, (0.2 + 2.8 * v), 0.0).astype(np.float32)
    # blobs
    pos = uniform01(seed * 7919 + 3, (n_blobs, 2)) * np.array([H - 1, W - 1])
    steps = (uniform01(seed * 7919 + 4, (T_all, n_blobs, 2)) * 2.0 - 1.0) * 10.0
    amp = 4.0 + 6.0 * uniform01(seed * 7919 + 5, (n_blobs,))
    r = 12
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
    for t in range(tb):
        pos = np.clip(pos + steps[t], 0, [H - 1, W - 1])
        if t < ta:
            continue
        for b in range(n_blobs):
            cy, cx = int(round(pos[b, 0])), int(round(pos[b, 1]))
            g = (amp[b] * np.exp(-(yy ** 2 + xx ** 2) / (2 * 4.0 ** 2))).astype(np.float32)
            g[g < 0.25] = 0
            y0, y1 = max(cy - r, 0), min(cy + r + 1, H)
            x0, x1 = max(cx - r, 0), min(cx + r + 1, W)
            sub = g[y0 - (cy - r):y1 - (cy - r), x0 - (cx - r):x1 - (cx - r)]
            frames[t - ta, y0:y1, x0:x1] = np.maximum(frames[t - ta, y0:y1, x0:x1], sub)
    return frames


def conv_layer_specs():
    """[(cin, cout, stride, pool_after)] for the 8 conv blocks of the deployed net."""
    specs, cin = [], IN_CHANNELS
    for i, a in enumerate(CONV_ARCH):
        if a == 'M':
            continue
        pool = (i + 1 < len(CONV_ARCH) and CONV_ARCH[i + 1] == 'M')
        specs.append((cin, a[1], a[2], pool))
        cin = a[1]
    return specs


def conv_block_names():
    return [f'ConvBlock_{i}' for i, a in enumerate(CONV_ARCH) if a != 'M']


def synth_state_dict(seed=42, conf_shift=0.05, gains=(2.0, 6.0, 1.0)):
    """Seeded weights with the reference's state_dict key set, as numpy f32 arrays.

    Conv weights ~ U(-b, b) with b = sqrt(6/fan_in); linear weights ~ U(-b, b) with
    b = gains[l]/sqrt(fan_in). BatchNorm running stats and affine parameters are non-trivial
    so that the BN fold is exercised. The last linear layer's biases are set so that roughly
    half of the 144 cells pass the 0.55 confidence floor (a few exceed 1.0, which exercises
    the 'scale_to_max' capping, AxonDetections.py:658-659) and the in-cell coordinates spread
    over about (-0.2, 1.0) with ~0.04 frame-to-frame jitter.
    """
    sd = {}
    s = int(seed) * 100
    for name, (ci, co, stride, pool) in zip(conv_block_names(), conv_layer_specs()):
        fan_in = ci * 9
        b = np.sqrt(6.0 / fan_in)             # He-uniform keeps activations O(1) through LeakyReLU
        sd[f'ConvNet.{name}.conv.weight'] = ((uniform01(s + 1, (co, ci, 3, 3)) * 2 - 1) * b).astype(np.float32)
        sd[f'ConvNet.{name}.conv.bias'] = ((uniform01(s + 2, (co,)) * 2 - 1) * 0.1).astype(np.float32)
        sd[f'ConvNet.{name}.batchnorm.weight'] = (0.75 + 0.5 * uniform01(s + 3, (co,))).astype(np.float32)
        sd[f'ConvNet.{name}.batchnorm.bias'] = (0.2 * normal(s + 4, (co,))).astype(np.float32)
        sd[f'ConvNet.{name}.batchnorm.running_mean'] = (0.2 * normal(s + 5, (co,))).astype(np.float32)
        sd[f'ConvNet.{name}.batchnorm.running_var'] = (0.5 + uniform01(s + 6, (co,))).astype(np.float32)
        sd[f'ConvNet.{name}.batchnorm.num_batches_tracked'] = np.array(1000, np.int64)
        s += 10
    feat = 160 * 16 * 16
    dims = [feat] + FC_ARCH + [SX * SY * 3]
    for k, idx in enumerate((1, 3, 5)):
        fi, fo = dims[k], dims[k + 1]
        b = gains[k] / np.sqrt(fi)
        sd[f'fcs.{idx}.weight'] = ((uniform01(s + 1, (fo, fi)) * 2 - 1) * b).astype(np.float32)
        sd[f'fcs.{idx}.bias'] = ((uniform01(s + 2, (fo,)) * 2 - 1) * 0.1).astype(np.float32)
        s += 10
    # output head: conf around the threshold, x/y around the cell centre
    bias = sd['fcs.5.bias'].reshape(SX * SY, 3).copy()
    w = sd['fcs.5.weight'].reshape(SX * SY, 3, -1)
    # sigmoid hidden units average 0.5 -> remove the mean drive, then add the target offset
    drive = 0.5 * w.sum(-1)
    bias[:, 0] += conf_shift + 0.55 - drive[:, 0]
    bias[:, 1] += 0.5 - drive[:, 1]
    bias[:, 2] += 0.5 - drive[:, 2]
    sd['fcs.5.bias'] = bias.reshape(-1).astype(np.float32)
    return sd


def corridor_mask(H, W, width=40, pitch=128):
    """Synthetic channel mask for BASELINE config 5: `width`-px corridors on a `pitch`-px lattice."""
    y = (np.arange(H) % pitch) < width
    x = (np.arange(W) % pitch) < width
    return (y[:, None] | x[None, :])
