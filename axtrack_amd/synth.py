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


def synth_detections(n_frames, H, W, n_alive=75, seed=0, mean_life=120, p_detect=0.92, clutter=0.08, max_step=10.0,
                     jitter=2.0, min_dist=23, cap=None):
    """Detections of a scene of MOVING growth cones, for association-only workloads (random walk of
    <= max_step px per frame, births and deaths) -- what the detector's output looks like on real timelapses, where the
    random-init benchmark weights give a static scene. About `n_alive` cones are alive in every frame; a cone lives
    ~mean_life frames (geometric), moves by a momentum random walk (reflecting at the borders), is missed with
    probability 1 - p_detect, is reported with +-jitter px of localisation noise and a confidence around its own level
    in [0.7, 1.1] (+-0.05; below the 0.55 floor = missed); `clutter` x n_alive false detections per frame with
    confidence in [0.55, 0.7). Every frame then goes through the detector's own greedy NMS rule (descending confidence,
    dx^2+dy^2 < min_dist^2 suppressed; AxonDetections.py:250-278) and is ordered by descending confidence.

    Returns dict(conf f32 [F,cap], x i32 [F,cap], y i32 [F,cap], count i32 [F], truth i32 [F,cap] (cone id, -1 = clutter)).
    Deterministic in (arguments, seed): raw PCG64 streams only."""
    F = int(n_frames)
    s = int(seed) * 15485863 + 11
    # cones: the initial population with random remaining lives, then births at the rate that keeps n_alive
    births = np.floor(uniform01(s + 1, (F,)) + n_alive / float(mean_life)).astype(np.int64)       # per frame, mean n_alive/mean_life
    births[0] = n_alive
    n_cones = int(births.sum())
    t_birth = np.repeat(np.arange(F), births)
    life = np.ceil(-np.log(1.0 - uniform01(s + 2, (n_cones,))) * mean_life).astype(np.int64).clip(1)
    level = 0.7 + 0.4 * uniform01(s + 3, (n_cones,))
    pos0 = uniform01(s + 4, (n_cones, 2)) * np.array([W - 1.0, H - 1.0])
    vel0 = (uniform01(s + 5, (n_cones, 2)) * 2 - 1) * max_step * 0.5
    cap = int(cap or max(64, -(-int(n_alive * (1.6 + clutter)) // 64) * 64))
    conf = np.zeros((F, cap), np.float32); x = np.zeros((F, cap), np.int32); y = np.zeros((F, cap), np.int32)
    truth = np.full((F, cap), -1, np.int32); count = np.zeros(F, np.int32)
    pos = pos0.copy(); vel = vel0.copy()
    lim = np.array([W - 1.0, H - 1.0])
    for t in range(F):
        alive = np.nonzero((t_birth <= t) & (t < t_birth + life))[0]
        u = uniform01(s + 100 + 7 * t, (n_cones, 6))
        # momentum random walk, speed capped at max_step px per frame (per axis), reflecting borders
        vel[alive] = np.clip(0.8 * vel[alive] + (u[alive, 0:2] * 2 - 1) * max_step * 0.4, -max_step, max_step)
        p = pos[alive] + vel[alive]
        over, under = p > lim, p < 0
        p = np.where(over, 2 * lim - p, np.where(under, -p, p))
        vel[alive] = np.where(over | under, -vel[alive], vel[alive])
        pos[alive] = p
        c = level[alive] + (u[alive, 2] - 0.5) * 0.1
        seen = (u[alive, 3] < p_detect) & (c >= 0.55)
        px = np.rint(p[seen, 0] + (u[alive, 4][seen] * 2 - 1) * jitter).clip(0, W - 1)
        py = np.rint(p[seen, 1] + (u[alive, 5][seen] * 2 - 1) * jitter).clip(0, H - 1)
        ids = alive[seen]
        nc = int(np.floor(uniform01(s + 101 + 7 * t, (1,))[0] + clutter * n_alive))
        uc = uniform01(s + 102 + 7 * t, (max(nc, 1), 3))[:nc]
        cc = np.concatenate([c[seen], 0.55 + 0.15 * uc[:, 0]]).astype(np.float32)
        cx = np.concatenate([px, np.floor(uc[:, 1] * W)]).astype(np.int64)
        cy = np.concatenate([py, np.floor(uc[:, 2] * H)]).astype(np.int64)
        ci = np.concatenate([ids, np.full(nc, -1)]).astype(np.int64)
        order = np.argsort(-cc, kind='stable')
        cc, cx, cy, ci = cc[order], cx[order], cy[order], ci[order]
        d2 = (cx[:, None] - cx[None]) ** 2 + (cy[:, None] - cy[None]) ** 2 < min_dist * min_dist
        keep = np.ones(len(cc), bool)
        for i in range(len(cc)):                               # greedy: a kept detection suppresses every later close one
            if keep[i]:
                keep[i + 1:] &= ~d2[i, i + 1:]
        n = min(int(keep.sum()), cap)
        count[t] = n
        conf[t, :n], x[t, :n], y[t, :n], truth[t, :n] = cc[keep][:n], cx[keep][:n], cy[keep][:n], ci[keep][:n]
    return dict(conf=conf, x=x, y=y, count=count, truth=truth)
