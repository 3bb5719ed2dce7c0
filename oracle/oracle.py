"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + the C library built from oracle/axt_oracle.c) of AxTrack's
detect+associate hot path. Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; axtrack_amd/ never does.

Every function cites the reference lines (relative to /root/reference) it follows. Pinning:
tests/test_oracle_golden.py checks this file against tests/golden/*.npz, produced by running
the reference's own Python (tests/golden/make_golden.py). pyastar2d and libmot/ortools are
absent from the reference tree -> `path_matrix` (masked case) and `mcf_solve` are restated
from the published algorithms, PARITY UNPINNED (conventions: DESIGN.md "Unpinned
third-party semantics").
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# hyper-parameters of the deployed model (deployed_model/params.txt) and the constants
# hard-coded in AxonDetections.__init__ (AxonDetections.py:76-78)
DEFAULTS = dict(
    TILESIZE=512, SX=12, SY=12, NON_MAX_SUPRESSION_DIST=23, BBOX_THRESHOLD=0.7,
    MCF_EDGE_COST_THR=0.7, MCF_ENTRY_EXIT_COST=2, MCF_MISS_RATE=0.6, MCF_MAX_NUM_MISSES=1,
    MCF_MIN_FLOW=5, MCF_MAX_FLOW=450, MCF_MAX_CONF_COST=4.6, MCF_VIS_SIM_WEIGHT=0,
    MCF_CONF_CAPPING_METHOD='scale_to_max',
)
CONF_FLOOR = np.float32(0.55)      # all_conf_thrs.min(), AxonDetections.py:76,122 (compared in f32)
MAX_PX_ASSOC_DIST = 500            # AxonDetections.py:77
AXON_BOX_SIZE = 70                 # AxonDetections.py:78
COST_SCALE = 1_000_000             # integer cost unit of the flow network (build convention)
PERT_BITS = 16                     # low bits: deterministic tie-breaking perturbation


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            subprocess.check_call(['make', '-C', _HERE, '-s'])
        _LIB = ctypes.CDLL(so)
        _LIB.orc_mcf_ssp.restype = ctypes.c_int
        _LIB.orc_astar_len.restype = ctypes.c_int
        # one thread per visible CPU is what the OpenMP runtime picks; a one-GPU box shows 256 CPUs and grants 16
        _LIB.orc_set_threads(min(len(os.sched_getaffinity(0)), 16))
    return _LIB


def set_threads(n):
    """Threads of the C restatement's parallel loops (OMP_NUM_THREADS is only read when the OpenMP runtime starts)."""
    lib().orc_set_threads(int(n))


def _p(a, t=ctypes.c_void_p):
    return a.ctypes.data_as(t)


# ------------------------------------------------------------------------------- a-1
def cnn_forward(sd, X, n_threads=None):
    """YOLO_AXTrack.detect_axons (model.py:50-53,119-125): X f32 [B,5,512,512] -> [B,12,12,3]."""
    from axtrack_amd import synth
    L = lib()
    if n_threads:
        set_threads(n_threads)
    x = np.ascontiguousarray(X, np.float32)
    B, C, H, W = x.shape
    for name, (ci, co, stride, pool) in zip(synth.conv_block_names(), synth.conv_layer_specs()):
        pre = f'ConvNet.{name}.'
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        out = np.empty((B, co, Ho, Wo), np.float32)
        arrs = [np.ascontiguousarray(sd[pre + k], np.float32) for k in
                ('conv.weight', 'conv.bias', 'batchnorm.weight', 'batchnorm.bias',
                 'batchnorm.running_mean', 'batchnorm.running_var')]
        L.orc_conv3x3_bn_lrelu(_p(x), B, ci, H, W, *[_p(a) for a in arrs], co, stride,
                               ctypes.c_float(0.1), _p(out))
        x, H, W = out, Ho, Wo
        if pool:
            out = np.empty((B, co, H // 2, W // 2), np.float32)
            L.orc_maxpool2(_p(x), B * co, H, W, _p(out))
            x, H, W = out, H // 2, W // 2
    x = x.reshape(B, -1)                                   # flatten(start_dim=1): (c, h, w) order
    for idx, sig in ((1, 1), (3, 1), (5, 0)):
        w = np.ascontiguousarray(sd[f'fcs.{idx}.weight'], np.float32)
        b = np.ascontiguousarray(sd[f'fcs.{idx}.bias'], np.float32)
        out = np.empty((B, w.shape[0]), np.float32)
        L.orc_linear(_p(x), B, w.shape[1], _p(w), _p(b), w.shape[0], sig, _p(out))
        x = out
    return x.reshape(B, 12, 12, 3)


# ------------------------------------------------------------------------------- f-1 (next row)
def preprocess(raw_u16, mask=None, offset=0.0, clip_lower=0.0, log_correct=True, scale=1.0):
    """Dense part of Timelapse._read_tiff/_clip_image_values/_log_adjust_image/_standardize
    (Timelapse.py:205-326) in numpy f32. img_as_float32 (x * (1/65535)) and adjust_log (log2(1+x)) are
    skimage functions absent here, restated from their published behaviour: PARITY UNPINNED."""
    x = np.multiply(raw_u16, 1. / 65535, dtype=np.float32)
    if mask is not None:
        m = np.asarray(mask, bool)
        if m.ndim == 2:
            x[:, ~m] = 0
        else:
            x[~m] = 0                                    # a mask per frame (Timelapse.py:214-217)
    if offset:
        x -= np.float32(offset)
        x[x < 0] = 0
    if clip_lower:
        x[x < np.float32(clip_lower)] = 0
    if log_correct:
        x = np.log2(1 + x / np.float32(1.0)) * np.float32(1.0)
    return (x / np.float32(scale)).astype(np.float32)


# ------------------------------------------------------------------------------- a-2 / tiling
def tile_grid(H, W, ts=512):
    return int(np.ceil(H / ts)), int(np.ceil(W / ts))


def kept_tiles(frames, ts=512):
    """Tiles that are non-empty (>0) at some t are kept, row-major (Timelapse.py:551-558)."""
    T, H, W = frames.shape
    ty, tx = tile_grid(H, W, ts)
    keep = []
    for iy in range(ty):
        for ix in range(tx):
            if (frames[:, iy * ts:(iy + 1) * ts, ix * ts:(ix + 1) * ts] > 0).any():
                keep.append((iy, ix))
    return keep


def frame_tile_stack(frames, t, keep, ts=512, ctx=2):
    """Timelapse.get_frametiles_stack(t) (Timelapse.py:111-125,150-157): [n_tiles,5,ts,ts];
    channel c = frame t+c of the context-padded sequence; edge tiles zero-padded (:529-533)."""
    T, H, W = frames.shape
    X = np.zeros((len(keep), 2 * ctx + 1, ts, ts), np.float32)
    for k, (iy, ix) in enumerate(keep):
        sub = frames[t:t + 2 * ctx + 1, iy * ts:(iy + 1) * ts, ix * ts:(ix + 1) * ts]
        X[k, :, :sub.shape[1], :sub.shape[2]] = sub
    return X


# ------------------------------------------------------------------------------- a-3 / a-4
def decode_filter(yolo, ts=512, S=12, thr=CONF_FLOOR):
    """_yolo_coo2tile_coo + _filter_yolo_det (AxonDetections.py:192-220) for a stack of tiles.

    yolo f32 [B,S,S,3], dim1 = x cell, dim2 = y cell. Returns per tile (conf f32, x i64,
    y i64, cell i64) in cell order (boolean-mask order, :220). f32 arithmetic, torch.round
    = half-to-even; cells that are all-zero stay zero (:194,209).
    """
    y = np.array(yolo, np.float32, copy=True)
    B = y.shape[0]
    zero = (y == 0).all(-1)
    ii = np.arange(S, dtype=np.float32).reshape(1, S, 1)
    jj = np.arange(S, dtype=np.float32).reshape(1, 1, S)
    xs = np.rint(((y[..., 1] + ii) * np.float32(ts)) / np.float32(S))
    ys = np.rint(((y[..., 2] + jj) * np.float32(ts)) / np.float32(S))
    xs[zero] = 0
    ys[zero] = 0
    out = []
    for b in range(B):
        conf = y[b, ..., 0].reshape(-1)
        m = conf >= thr
        cell = np.nonzero(m)[0]
        out.append((conf[m], xs[b].reshape(-1)[m].astype(np.int64), ys[b].reshape(-1)[m].astype(np.int64), cell))
    return out


# ------------------------------------------------------------------------------- a-5
def stitch(tile_dets, keep, ts=512):
    """Timelapse.stitch_tiles (Timelapse.py:166-197): tile -> frame coordinates, concatenated
    in kept-tile order, each tile's table in ascending-conf order (_torch2pandas, :235)."""
    conf, x, y, key = [], [], [], []
    for k, ((c, ax, ay, cell), (iy, ix)) in enumerate(zip(tile_dets, keep)):
        order = np.lexsort((cell, c))                  # conf ascending, ties in cell order
        conf.append(c[order]); x.append(ax[order] + ix * ts); y.append(ay[order] + iy * ts)
    if not conf:
        return np.zeros(0, np.float32), np.zeros(0, np.int64), np.zeros(0, np.int64)
    return np.concatenate(conf), np.concatenate(x), np.concatenate(y)


# ------------------------------------------------------------------------------- a-6
def nms(conf, x, y, min_dist=23):
    """_non_max_supression (AxonDetections.py:250-278): descending conf (ties keep the
    concatenation order), each survivor drops every later row with
    sqrt(int(dx^2+dy^2)) < min_dist, i.e. dx^2+dy^2 < min_dist^2 (strict)."""
    order = np.argsort(-conf.astype(np.float64), kind='stable')
    conf, x, y = conf[order], x[order], y[order]
    alive = np.ones(len(conf), bool)
    thr2 = int(min_dist) * int(min_dist)
    for i in range(len(conf)):
        if not alive[i]:
            continue
        d2 = (x - x[i]) ** 2 + (y - y[i]) ** 2
        kill = d2 < thr2
        kill[:i + 1] = False
        alive &= ~kill
    return conf[alive], x[alive], y[alive]


def detect_from_yolo(yolo_frames, keep, ts=512, min_dist=23):
    """decode -> filter -> stitch -> NMS for a list of per-frame YOLO stacks."""
    dets = []
    for yolo in yolo_frames:
        dets.append(nms(*stitch(decode_filter(yolo, ts), keep, ts), min_dist))
    return dets


def detect_dataset(frames, sd, ts=512, min_dist=23, return_yolo=False):
    """AxonDetections.detect_dataset (AxonDetections.py:87-139) on dense frames [T_all,H,W]."""
    keep = kept_tiles(frames, ts)
    n_frames = frames.shape[0] - 4
    yolo = [cnn_forward(sd, frame_tile_stack(frames, t, keep, ts)) for t in range(n_frames)]
    dets = detect_from_yolo(yolo, keep, ts, min_dist)
    return (dets, yolo) if return_yolo else dets


# ------------------------------------------------------------------------------- a-7 / a-8
def libmot_rows(dets):
    """get_frame_dets('all', None, libmot=True) (AxonDetections.py:313-317,754-784):
    rows [frame, id, x-35, y-35, 70, 70, conf(float64 of the f32 value)]."""
    rows = []
    for t, (c, x, y) in enumerate(dets):
        for i in range(len(c)):
            rows.append((t, i, x[i] - AXON_BOX_SIZE // 2, y[i] - AXON_BOX_SIZE // 2,
                         AXON_BOX_SIZE, AXON_BOX_SIZE, float(c[i])))
    return np.array(rows, np.float64).reshape(-1, 7)


def cap_conf(conf64, method='scale_to_max'):
    """AxonDetections.py:655-659 (python-float, i.e. f64, arithmetic)."""
    c = np.array(conf64, np.float64, copy=True)
    if method == 'ceil':
        c[c > 1] = 1
    if method == 'scale_to_max':
        c = c / c.max()
    return c


# ------------------------------------------------------------------------------- a-10 / a-11
def observation_cost(scores, max_conf_cost=4.6):
    """observation_model (mincostflow_models.py:6-27), f64."""
    s = np.asarray(scores, np.float64)
    s = (s - 1) * -1 + 1e-6
    s = np.log(s / (1 - s))
    s[s > max_conf_cost] = max_conf_cost
    s[s < -max_conf_cost] = -max_conf_cost
    return s


def transition_cost(D, gap, miss_rate=0.6, max_px=MAX_PX_ASSOC_DIST, vis_w=0, vis_sim=None):
    """transition_model (mincostflow_models.py:67-119), f64. vis_sim: matrix of 1 - Bhattacharyya distance of
    the appearance features (needed iff vis_w > 0; nan_to_num is a no-op on it, the distance is never nan)."""
    distances = ((np.asarray(D) / max_px) - 1) * -1
    inf = distances == 0
    vs = 0.0 if vis_w == 0 else np.asarray(vis_sim, np.float64)
    with np.errstate(divide='ignore'):
        costs = -np.log((1 - vis_w) * distances * (miss_rate ** (gap - 1)) + vis_w * vs + 1e-6)
    costs = np.asarray(costs, np.float64)
    costs[inf] = np.inf
    return costs


# ------------------------------------------------------------------------------- f-3 (next row)
def box_histograms(image, x, y, box=AXON_BOX_SIZE):
    """feature_model (mincostflow_models.py:30-65) for the libmot boxes (x - box//2, y - box//2, box, box) that
    det2libmot_det builds (AxonDetections.py:754-784): f32 [n,180]. cv2 restated, PARITY UNPINNED (axt_oracle.c)."""
    img = np.ascontiguousarray(image, np.float32)
    xs, ys = np.ascontiguousarray(x, np.int64), np.ascontiguousarray(y, np.int64)
    out = np.zeros((len(xs), 180), np.float32)
    if len(xs):
        lib().orc_box_histograms(_p(img), img.shape[0], img.shape[1], _p(xs), _p(ys), len(xs), int(box), _p(out))
    return out


def bhattacharyya(h1, h2):
    """cv2.compareHist(f1, f2, HISTCMP_BHATTACHARYYA) for every pair (mincostflow_models.py:107-113): f64 [n1,n2]."""
    a, b = np.ascontiguousarray(h1, np.float32), np.ascontiguousarray(h2, np.float32)
    out = np.zeros((len(a), len(b)), np.float64)
    if len(a) and len(b):
        lib().orc_bhattacharyya(_p(a), len(a), _p(b), len(b), _p(out))
    return out


# ------------------------------------------------------------------------------- a-9
def path_matrix(src, dst, H, W, mask=None, max_dist=MAX_PX_ASSOC_DIST, conn8=False):
    """_compute_detections_astar_paths + _get_astar_path_distances for one frame pair
    (AxonDetections.py:526-629,717-752): D[i,j] = number of cells of the A* path between
    detection i at t_bef (src) and j at t (dst) on weights {1 on mask, 65536 off} (:598);
    max_dist where euclid >= max_dist or no path within max_dist cells.
    mask=None means all-ones: closed form |dx|+|dy|+1 (4-connected). An end point outside
    the H x W grid has no path (upstream pyastar2d rejects it): max_dist."""
    (_, sx, sy), (_, dx_, dy_) = src, dst
    ns, nd = len(sx), len(dx_)
    if ns == 0:
        return np.zeros((0,), np.int32)          # np.array([]) in the reference (:747)
    if mask is None:
        ddx = np.abs(sx[:, None] - dx_[None, :]); ddy = np.abs(sy[:, None] - dy_[None, :])
        eu = np.sqrt((ddx ** 2 + ddy ** 2).astype(np.float64))
        L = (np.maximum(ddx, ddy) if conn8 else ddx + ddy) + 1
        inb = (((sx >= 0) & (sx < W) & (sy >= 0) & (sy < H))[:, None]
               & ((dx_ >= 0) & (dx_ < W) & (dy_ >= 0) & (dy_ < H))[None, :])
        return np.where((eu < max_dist) & (L <= max_dist) & inb, L, max_dist).astype(np.int32)
    weights = np.ascontiguousarray(np.where(mask == 1, 1, 2 ** 16).astype(np.float32))
    Hh, Ww = weights.shape
    D = np.empty((ns, nd), np.int32)
    a = [np.ascontiguousarray(v, np.int64) for v in (sx, sy, dx_, dy_)]
    lib().orc_path_matrix(_p(weights), Hh, Ww, _p(a[0]), _p(a[1]), ns, _p(a[2]), _p(a[3]), nd,
                          int(max_dist), int(bool(conn8)), _p(D))
    return D


def mask_of_frame(mask, t, quirk=True, ctx=2):
    """The mask the reference searches paths on for the pairs that END in detection frame t (_get_maskweights(t),
    AxonDetections.py:557,587-598). A [H,W] mask holds for every frame (Timelapse.py:214-215). A [T_all,H,W] mask is
    indexed with the DETECTION frame index t although dataset.mask lists the context-padded input frames
    (Timelapse.py:408), i.e. it takes the mask of input frame t, not of t + ctx where detection frame t sits --
    reproduced by default (quirk=True)."""
    if mask is None:
        return None
    mask = np.asarray(mask)
    if mask.ndim == 2:
        return mask
    return mask[t if quirk else t + ctx]


def all_path_matrices(dets, H, W, mask=None, max_misses=1, name='synth', mask_quirk=True, **kw):
    """dict keyed like the reference: '{name}_t:{t:03}-t:{t_bef:03}' (:554,563). mask: None, [H,W] or [T_all,H,W]."""
    out = {}
    for t in range(len(dets)):
        m = mask_of_frame(mask, t, mask_quirk)
        for t_bef in range(t - 1, t - (max_misses + 2), -1):
            if t_bef < 0:
                continue
            out[f'{name}_t:{t:0>3}-t:{t_bef:0>3}'] = path_matrix(dets[t_bef], dets[t], H, W, m, **kw)
    return out


# ------------------------------------------------------------------------------- a-12
def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def arc_cost_int(cost, kind, a, b):
    """Integer arc cost = round(cost * COST_SCALE) << PERT_BITS | perturbation(kind, a, b).

    kind: 0 entry (a = global det index), 1 exit, 2 observation, 3 transition (a -> b).
    The low PERT_BITS bits are a hash of the arc's identity: it makes the optimum unique, so
    every exact solver returns the same tracks (build convention; libmot is unpinned)."""
    key = (int(kind) << 60) ^ (int(a) << 30) ^ int(b)
    pert = _splitmix64(key) & ((1 << PERT_BITS) - 1)
    return (int(np.rint(cost * COST_SCALE)) << PERT_BITS) + pert


def build_flow_graph(dets, D, P=DEFAULTS, name='synth', images=None):
    """Arc lists of the tracking network the reference hands to libmot (call sites
    AxonDetections.py:663-690): node 0 = source, 1 = sink, detection k -> (2+2k, 3+2k).
    images: per detection frame the image feature_model sees (the stitched centre frame, :682-685); needed iff
    MCF_VIS_SIM_WEIGHT > 0."""
    counts = [len(d[0]) for d in dets]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    conf = np.concatenate([np.asarray(d[0], np.float32) for d in dets]).astype(np.float64) if sum(counts) else np.zeros(0)
    capped = cap_conf(conf, P['MCF_CONF_CAPPING_METHOD']) if len(conf) else conf
    obs = observation_cost(capped, P['MCF_MAX_CONF_COST']) if len(conf) else conf
    tail, head, cost = [], [], []
    ee = float(P['MCF_ENTRY_EXIT_COST'])
    vis_w = P.get('MCF_VIS_SIM_WEIGHT', 0)
    feats = [box_histograms(images[t], d[1], d[2]) for t, d in enumerate(dets)] if vis_w else None
    for k in range(int(offs[-1])):
        tail.append(0); head.append(2 + 2 * k); cost.append(arc_cost_int(ee, 0, k, 0))
        tail.append(2 + 2 * k); head.append(3 + 2 * k); cost.append(arc_cost_int(obs[k], 2, k, 0))
        tail.append(3 + 2 * k); head.append(1); cost.append(arc_cost_int(ee, 1, k, 0))
    for t in range(len(dets)):
        for gap in range(1, P['MCF_MAX_NUM_MISSES'] + 2):
            tb = t - gap
            if tb < 0 or counts[t] == 0 or counts[tb] == 0:
                continue
            Dm = D[f'{name}_t:{t:0>3}-t:{tb:0>3}']
            vs = 1 - bhattacharyya(feats[tb], feats[t]) if vis_w else None
            c = transition_cost(Dm, gap, P['MCF_MISS_RATE'], vis_w=vis_w, vis_sim=vs)
            ii, jj = np.nonzero(c < P['MCF_EDGE_COST_THR'])
            for i, j in zip(ii, jj):
                a, b = int(offs[tb] + i), int(offs[t] + j)
                tail.append(3 + 2 * a); head.append(2 + 2 * b); cost.append(arc_cost_int(c[i, j], 3, a, b))
    return (np.array(tail, np.int32), np.array(head, np.int32), np.array(cost, np.int64), offs)


def mcf_solve(dets, D, P=DEFAULTS, name='synth', images=None):
    """MinCostFlowTracker.process x frames + compute_trajectories (AxonDetections.py:679-690).

    Returns a list of trajectories, each a list of (frame, det_idx), or None if fewer than
    MCF_MIN_FLOW unit flows exist (the reference's falsy result, :691). Canonical order:
    by (first frame, det idx of the first detection)."""
    tail, head, cost, offs = build_flow_graph(dets, D, P, name, images)
    n_det = int(offs[-1])
    n_nodes = 2 + 2 * n_det
    flow = np.zeros(len(tail), np.uint8)
    tot = ctypes.c_int64(0)
    feas = ctypes.c_int(0)
    F = lib().orc_mcf_ssp(n_nodes, len(tail), _p(tail), _p(head), _p(cost),
                          int(P['MCF_MIN_FLOW']), int(P['MCF_MAX_FLOW']), _p(flow),
                          ctypes.byref(tot), ctypes.byref(feas))
    if not feas.value or F == 0:
        return None, int(tot.value)
    nxt = {}
    starts = []
    for a in np.nonzero(flow)[0]:
        tl, hd = int(tail[a]), int(head[a])
        if tl == 0:
            starts.append((hd - 2) // 2)
        elif hd != 1 and tl % 2 == 1 and hd % 2 == 0:
            nxt[(tl - 3) // 2] = (hd - 2) // 2
    frame_of = np.searchsorted(offs, np.arange(n_det), side='right') - 1
    trajs = []
    for k in sorted(starts):
        tr = []
        while k is not None:
            f = int(frame_of[k])
            tr.append((f, int(k - offs[f])))
            k = nxt.get(k)
        trajs.append(tr)
    return trajs, int(tot.value)


# ------------------------------------------------------------------------------- config 3 variant
def hungarian_assoc(dets, H, W, P=DEFAULTS, mask=None, images=None):
    """Frame-to-frame Hungarian association (BASELINE config 3; the reference has no such code,
    SURVEY.md F7 -- this is the build's own definition, solved here with SciPy's
    linear_sum_assignment as the independent exact solver).

    Pass 1, every pair (t, t+1): rows = detections of t, columns = detections of t+1 plus one
    private "no successor" column per row; cost = integer transition cost (arc_cost_int kind 3) where
    transition_cost < MCF_EDGE_COST_THR, forbidden otherwise; the private column costs
    arc_cost_int(THR, 1, a, 0). Pass 2, pairs (t, t+2), restricted to rows without successor and
    columns without predecessor after pass 1. Chains are numbered by (first frame, index).
    images: per detection frame the image feature_model sees; needed iff MCF_VIS_SIM_WEIGHT > 0 (the transition cost then
    carries the appearance term exactly as in build_flow_graph).
    Returns the list of trajectories [(frame, idx), ...] in id order."""
    from scipy.optimize import linear_sum_assignment
    counts = [len(d[0]) for d in dets]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    F = len(dets)
    thr = P['MCF_EDGE_COST_THR']
    vis_w = P.get('MCF_VIS_SIM_WEIGHT', 0)
    feats = [box_histograms(images[t], d[1], d[2]) for t, d in enumerate(dets)] if vis_w else None
    succ = [np.full(n, -1, np.int64) for n in counts]       # (frame offset encoded separately)
    succ_gap = [np.zeros(n, np.int64) for n in counts]
    has_pred = [np.zeros(n, bool) for n in counts]

    def solve(t, gap, rows, cols):
        tb = t + gap
        if len(rows) == 0:
            return
        sub_s = tuple(a[rows] for a in dets[t])
        sub_d = tuple(a[cols] for a in dets[tb])
        m_pair = mask_of_frame(mask, tb, P.get('REPRODUCE_MASK_FRAME_QUIRK', True))        # the mask of the pair's later frame
        D = path_matrix(sub_s, sub_d, H, W, m_pair) if len(cols) else np.zeros((len(rows), 0), np.int32)
        vs = (1 - bhattacharyya(feats[t][rows], feats[tb][cols])) if (vis_w and len(cols)) else None
        c = transition_cost(D, gap, P['MCF_MISS_RATE'], vis_w=vis_w if len(cols) else 0, vis_sim=vs)
        n, m = len(rows), len(cols)
        M = np.full((n, m + n), np.inf)
        for i in range(n):
            a = int(offs[t] + rows[i])
            for j in np.nonzero(c[i] < thr)[0]:
                M[i, j] = float(arc_cost_int(c[i, j], 3, a, int(offs[tb] + cols[j])))
            M[i, m + i] = float(arc_cost_int(thr, 1, a, 0))
        ri, ci = linear_sum_assignment(M)
        for i, j in zip(ri, ci):
            if j < m:
                succ[t][rows[i]] = cols[j]
                succ_gap[t][rows[i]] = gap
                has_pred[tb][cols[j]] = True

    for t in range(F - 1):
        solve(t, 1, np.arange(counts[t]), np.arange(counts[t + 1]))
    if P['MCF_MAX_NUM_MISSES'] >= 1:
        pred1 = [h.copy() for h in has_pred]
        for t in range(F - 2):
            solve(t, 2, np.nonzero(succ[t] < 0)[0], np.nonzero(~pred1[t + 2])[0])
    trajs = []
    for t in range(F):
        for i in range(counts[t]):
            if has_pred[t][i]:
                continue
            tr, f, k = [], t, i
            while True:
                tr.append((f, int(k)))
                if succ[f][k] < 0:
                    break
                f, k = f + int(succ_gap[f][k]), succ[f][k]
            trajs.append(tr)
    return trajs


# ------------------------------------------------------------------------------- a-13
def ided_tables(trajs, dets):
    """:699-711 + libmot_det2det (:786-823): per frame, rows (id, conf, x, y) sorted by id.
    conf is re-looked-up from the UNCAPPED detections by exact (x, y) match (:804-808); NMS
    guarantees anchors are unique within a frame."""
    per = [[] for _ in dets]
    for tid, tr in enumerate(trajs):
        for f, k in tr:
            c, x, y = dets[f]
            per[f].append((tid, float(c[k]), int(x[k]), int(y[k])))
    return [sorted(p) for p in per]


def ided_dets_all(tables, reproduce_label_quirk=True):
    """_agg_all_IDed_dets (AxonDetections.py:825-842) as plain arrays:
    returns (ids sorted, frame_labels[3F], info[3F], values[n_ids, 3F] f64 with NaN).

    Quirk reproduced by default: column labels are position//3 (:833), so after a frame with
    no IDed detection later frames are labelled one too low and the trailing labels are
    NaN-filled (:836-839)."""
    nF = len(tables)
    ids = sorted({r[0] for p in tables for r in p})
    row = {i: n for n, i in enumerate(ids)}
    blocks = []
    for f, p in enumerate(tables):
        if not p and reproduce_label_quirk:
            continue
        blk = np.full((len(ids), 3), np.nan)
        for tid, c, x, y in p:
            blk[row[tid]] = (x, y, c)          # columns sorted: anchor_x, anchor_y, conf
        blocks.append(blk)
    while len(blocks) < nF:
        blocks.append(np.full((len(ids), 3), np.nan))
    vals = np.concatenate(blocks, 1) if blocks else np.zeros((len(ids), 0))
    labels = np.repeat(np.arange(nF), 3).astype(np.float64)
    info = np.tile(np.array(['anchor_x', 'anchor_y', 'conf']), nF)
    return ids, labels, info, vals


# ------------------------------------------------------------------------------- whole path
# ------------------------------------------------------------------------------- f-4 (next row)
def all_conf_thrs(bbox_thr=0.7):
    """AxonDetections.py:76"""
    return np.sort(np.append(np.arange(0.55, 1, .04), bbox_thr)).round(2)


def detection_confusion(det, gt_x, gt_y, thrs=None, min_dist=23, return_masks_at=None):
    """compute_TP_FP_FN (AxonDetections.py:409-466) for one frame: int [3 (TP, FP, FN), len(thrs)].
    det = (conf f32, x, y) in descending confidence; ground truth anchors gt_x, gt_y. Restated with its quirks:
      * an empty side is replaced by ONE row (conf, x, y) = (0, 0, 0) (:434-437) -- a phantom label / detection at
        the origin that takes part in the matching;
      * per label, the candidates are the detections closer than min_dist (sqrt(dx^2+dy^2) < 23, exact for integer
        anchors) with conf > thr (f32 conf against the f64 threshold, compared in f64 as numpy >= 2 does); the
        closest wins, ties to the first;
      * labels are visited in order; a label whose closest candidate was already claimed by an earlier label is a
        false negative -- it does NOT fall back to its second-closest candidate (:447-452).
    return_masks_at: threshold index -> (FP mask over detections, FN mask over labels) instead (:462-465)."""
    thrs = all_conf_thrs() if thrs is None else np.asarray(thrs, np.float64)
    conf, x, y = (np.asarray(v) for v in det)
    if len(conf) == 0:
        conf, x, y = np.zeros(1, np.float32), np.zeros(1, np.int64), np.zeros(1, np.int64)
    gt_x, gt_y = np.asarray(gt_x, np.int64), np.asarray(gt_y, np.int64)
    if len(gt_x) == 0:
        gt_x, gt_y = np.zeros(1, np.int64), np.zeros(1, np.int64)
    d2 = (gt_x[:, None] - np.asarray(x, np.int64)[None]) ** 2 + (gt_y[:, None] - np.asarray(y, np.int64)[None]) ** 2
    c64 = np.asarray(conf, np.float32).astype(np.float64)
    out = np.zeros((3, len(thrs)), np.int64)
    for k, thr in enumerate(thrs):
        above = c64 > thr
        tp = np.zeros(len(c64), bool)
        taken = []
        fn = np.zeros(len(gt_x), bool)
        for i in range(len(gt_x)):
            cand = np.nonzero((d2[i] < min_dist * min_dist) & above)[0]
            best = int(cand[np.argmin(d2[i][cand])]) if len(cand) else -1
            if best >= 0 and best not in taken:
                taken.append(best)
            else:
                fn[i] = True
        tp[taken] = True
        fp = ~tp & above
        out[:, k] = tp.sum(), fp.sum(), fn.sum()
        if return_masks_at is not None and k == return_masks_at:
            return fp, fn
    return out


def prc_rcl_f1(cm):
    """compute_prc_rcl_F1 (AxonDetections.py:468-503)."""
    prc = cm[0] / (cm[0] + cm[1] + 1e-6)
    rcl = cm[0] / (cm[0] + cm[2] + 1e-6)
    f1 = 2 * (prc * rcl) / ((prc + rcl) + 1e-6)
    return np.array([prc, rcl, f1]).round(3)


def inference(frames, sd, mask=None, P=DEFAULTS, name='synth', yolo=None, assoc='mcf'):
    """interface.inference (interface.py:170-215): detect_dataset + assign_ids.
    assoc='mcf' is the reference's global tracker; 'hungarian' the BASELINE config 3 variant."""
    keep = kept_tiles(frames, P['TILESIZE'])
    if yolo is None:
        dets, yolo = detect_dataset(frames, sd, P['TILESIZE'], P['NON_MAX_SUPRESSION_DIST'], return_yolo=True)
    else:
        dets = detect_from_yolo(yolo, keep, P['TILESIZE'], P['NON_MAX_SUPRESSION_DIST'])
    if assoc == 'hungarian':
        images = [frames[t + 2] for t in range(len(dets))] if P.get('MCF_VIS_SIM_WEIGHT', 0) else None
        trajs = hungarian_assoc(dets, frames.shape[1], frames.shape[2], P, mask, images)
        tables = ided_tables(trajs, dets)
        return dict(dets=dets, yolo=yolo, D=None, trajs=trajs, total_cost=None, tables=tables,
                    ided_all=ided_dets_all(tables))
    D = all_path_matrices(dets, frames.shape[1], frames.shape[2], mask, P['MCF_MAX_NUM_MISSES'], name,
                          mask_quirk=P.get('REPRODUCE_MASK_FRAME_QUIRK', True))
    images = [frames[t + 2] for t in range(len(dets))] if P.get('MCF_VIS_SIM_WEIGHT', 0) else None   # the centre frames
    trajs, total = mcf_solve(dets, D, P, name, images)
    tables = ided_tables(trajs, dets) if trajs else None
    return dict(dets=dets, yolo=yolo, D=D, trajs=trajs, total_cost=total, tables=tables,
                ided_all=ided_dets_all(tables) if tables else None)
