"""Thin Python wrappers over the C ABI (include/axtrack_hip.h): torch tensors own the device
memory and the stream, the HIP library does the work. Each wrapper names the reference call
it stands in for."""
import ctypes
import os

import numpy as np
import torch

from . import _lib

TILE, S, CELLS = 512, 12, 144
CONF_FLOOR = float(np.float32(0.55))   # all_conf_thrs.min() as f32 (AxonDetections.py:76,122)
MAX_PX_ASSOC_DIST = 500                # AxonDetections.py:77
AXON_BOX_SIZE = 70                     # AxonDetections.py:78


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu():
    if not torch.cuda.is_available():
        raise _lib.AxtError('axtrack_amd needs a ROCm GPU (MI355X, gfx950); there is no CPU path')


# ConvBlock_i modules of the deployed ARCHITECTURE that hold a convolution (blocks 3, 6, 9 are the max-pools;
# deployed_model/params.txt:34, model.py:85-103)
CONV_BLOCKS = (0, 1, 2, 4, 5, 7, 8, 10)


def state_dict_tensor_order():
    """Key order axt_detector_create expects (include/axtrack_hip.h)."""
    keys = []
    for name in (f'ConvBlock_{i}' for i in CONV_BLOCKS):
        for k in ('conv.weight', 'conv.bias', 'batchnorm.weight', 'batchnorm.bias',
                  'batchnorm.running_mean', 'batchnorm.running_var'):
            keys.append(f'ConvNet.{name}.{k}')
    for idx in (1, 3, 5):
        keys += [f'fcs.{idx}.weight', f'fcs.{idx}.bias']
    return keys


class Detector:
    """Device-resident YOLO_AXTrack (model.py:20-125) in eval mode; `detect_axons` keeps the
    reference's name and tensor contract (model.py:119-125)."""

    ARITH = {'f32': 2, 'f32_winograd': 2, 'f32_direct': 0, 'bf16x3': 1}

    def __init__(self, state_dict, max_batch=256, device='cuda:0', arith='f32'):
        _require_gpu()
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        lib = _lib.load()
        keep = []
        for k in state_dict_tensor_order():
            if k not in state_dict:
                raise KeyError(f'state_dict lacks {k}')
            v = state_dict[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            keep.append(np.ascontiguousarray(v, np.float32))
        arr = (ctypes.c_void_p * len(keep))(*[a.ctypes.data for a in keep])
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.axt_detector_create(arr, len(keep), self.max_batch, ctypes.byref(handle)),
                       'axt_detector_create')
        self._h = handle
        self._lib = lib
        self.arith = 'f32'              # axt_detector_create leaves the handle in its default mode (f32 Winograd)
        self.fused_front = os.environ.get('AXT_FUSE_S2', '1') != '0'     # what axt_detector_create read
        self.set_arith(arith)

    def set_arith(self, arith):
        """Arithmetic of the stride-1 conv blocks (parameters['CNN_ARITH']): 'f32' (default; the
        same as 'f32_winograd': Winograd F(2x2,3x3) on the f32 matrix pipe, every operation f32), 'f32_direct' (direct
        convolution on the f32 matrix pipe: a k-ordered chain of f32 FMAs) or 'bf16x3' (opt-in: operands split into three
        bf16 terms, six partial products on the bf16 matrix pipe, f32 accumulation). See axt_detector_set_arith."""
        if arith not in self.ARITH:
            raise ValueError(f"CNN_ARITH must be one of {sorted(self.ARITH)}, got {arith!r}")
        if self.ARITH[arith] != self.ARITH[self.arith]:
            with torch.cuda.device(self.device):
                _lib.check(self._lib.axt_detector_set_arith(self._h, self.ARITH[arith]), 'axt_detector_set_arith')
        self.arith = arith

    def set_fused_front(self, fused):
        """The two stride-2 conv blocks as one kernel that keeps block 0's output in LDS (default) or as the two separate
        kernels (axt_detector_set_fused_front): the grids agree to f32 rounding, not bit for bit."""
        _lib.check(self._lib.axt_detector_set_fused_front(self._h, 1 if fused else 0), 'axt_detector_set_fused_front')
        self.fused_front = bool(fused)

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._lib.axt_detector_destroy(h)

    @property
    def device_bytes(self):
        return int(self._lib.axt_detector_device_bytes(self._h))

    KERNEL_NAMES = ['conv0 5>20 s2', 'conv1 20>40 s2', 'conv2 40>80 +pool', 'conv4 80>80', 'conv5 80>80 +pool',
                    'conv7 80>80', 'conv8 80>80 +pool', 'conv10 80>160', 'fc1 gemm', 'fc1 reduce', 'fc2 gemm',
                    'fc2 reduce', 'fc3 gemm', 'fc3 reduce']

    def set_profiling(self, on, only=None):
        """Bracket the forward pass's kernel launches with HIP events: all of them, or only kernel index `only`."""
        mode = 0 if not on else (1 if only is None else 2 + int(only))
        _lib.check(self._lib.axt_detector_set_profiling(self._h, mode), 'axt_detector_set_profiling')

    def read_profile(self):
        """Per-kernel HIP-event times since the last read: list of dicts (name, ms, launches, tiles, flops_per_tile)."""
        n = len(self.KERNEL_NAMES)
        ms = np.zeros(n, np.float64)
        launches = np.zeros(n, np.int64)
        items = np.zeros(n, np.int64)
        _lib.check(self._lib.axt_detector_read_profile(self._h, ms.ctypes.data, launches.ctypes.data,
                                                       items.ctypes.data, n), 'axt_detector_read_profile')
        rows = [dict(name=self.KERNEL_NAMES[i], ms=float(ms[i]), launches=int(launches[i]), tiles=int(items[i]),
                     flops_per_tile=float(self._lib.axt_cnn_kernel_flops_per_tile(i))) for i in range(n)]
        if rows[0]['launches'] and not rows[1]['launches']:
            # the fused front kernel is bracketed as kernel 0 and does the work of kernels 0 and 1 (row 1 stays, empty)
            rows[0]['name'] = 'conv0+1 5>20>40 s2 fused'
            rows[0]['flops_per_tile'] += rows[1]['flops_per_tile']
        return rows

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise _lib.AxtError('the detector is bound to the device it was created on')
        return self

    def detect_axons(self, X):
        """X f32 [B,5,512,512] on the GPU -> [B,12,12,3] (model.py:119-125)."""
        if X.dim() != 4 or tuple(X.shape[1:]) != (5, TILE, TILE):
            raise ValueError(f'expected [B,5,{TILE},{TILE}], got {tuple(X.shape)}')
        X = X.to(self.device, torch.float32).contiguous()
        out = torch.empty((X.shape[0], S, S, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.axt_cnn_forward(self._h, X.data_ptr(), X.shape[0], out.data_ptr(), _stream()),
                       'axt_cnn_forward')
        return out

    def detect_frames(self, frames, tile_yx, t0=0, n_frames=None):
        """frames f32 [T_all,H,W] on the GPU -> YOLO grids [n_frames, n_tiles, 12,12,3] for detection
        frames t0..t0+n_frames-1 (get_frametiles_stack + detect_axons, AxonDetections.py:115-118)."""
        T_all, H, W = frames.shape
        if n_frames is None:
            n_frames = T_all - 4 - t0
        tile_yx = np.ascontiguousarray(tile_yx, np.int32).reshape(-1, 2)
        out = torch.empty((n_frames, len(tile_yx), S, S, 3), dtype=torch.float32, device=self.device)
        assert frames.is_contiguous() and frames.dtype == torch.float32 and frames.device == self.device
        with torch.cuda.device(self.device):
            _lib.check(self._lib.axt_cnn_forward_frames(self._h, frames.data_ptr(), T_all, H, W, t0, n_frames,
                                                        tile_yx.ctypes.data, len(tile_yx), out.data_ptr(), _stream()),
                       'axt_cnn_forward_frames')
        return out


    def front_frames(self, frames, tile_yx, t0, n_frames, item0):
        """Conv blocks 0-5 of detection frames t0 .. t0+n_frames-1 into the detector's batch buffer from item `item0`
        (axt_cnn_front_frames): input that arrives in chunks. back() finishes the pass for all items at once."""
        T_all, H, W = frames.shape
        tile_yx = np.ascontiguousarray(tile_yx, np.int32).reshape(-1, 2)
        assert frames.is_contiguous() and frames.dtype == torch.float32 and frames.device == self.device
        with torch.cuda.device(self.device):
            _lib.check(self._lib.axt_cnn_front_frames(self._h, frames.data_ptr(), T_all, H, W, t0, n_frames, tile_yx.ctypes.data,
                                                      len(tile_yx), int(item0), _stream()), 'axt_cnn_front_frames')

    def back(self, n_frames, n_tiles):
        """The rest of the forward pass for items 0 .. n_frames*n_tiles-1 of the batch buffer -> [n_frames, n_tiles, 12,12,3]."""
        out = torch.empty((n_frames, n_tiles, S, S, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.axt_cnn_back(self._h, n_frames * n_tiles, out.data_ptr(), _stream()), 'axt_cnn_back')
        return out


def preprocess_u16(raw, mask=None, offset=0.0, clip_lower=0.0, log_correct=True, scale=1.0, out=None):
    """Fused dense preprocessing (Timelapse.py:205-326) of a raw uint16 timelapse that already sits on the GPU.
    raw: torch.uint16 (or int16-viewed) [T,H,W]; mask: uint8/bool [H,W] on the same device or None; out: optional
    contiguous f32 [T,H,W] to write into (a slice of the timelapse's frame buffer when the input arrives in chunks)."""
    T, H, W = raw.shape
    assert raw.is_contiguous() and raw.element_size() == 2
    if out is None:
        out = torch.empty((T, H, W), dtype=torch.float32, device=raw.device)
    assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (T, H, W) and out.device == raw.device
    if mask is not None:
        mask = mask.to(device=raw.device, dtype=torch.uint8).contiguous()
    lib = _lib.load()
    with torch.cuda.device(raw.device):
        _lib.check(lib.axt_preprocess_u16(raw.data_ptr(), _lib.dptr(mask), T, H, W, ctypes.c_float(offset),
                                          ctypes.c_float(clip_lower), int(bool(log_correct)), ctypes.c_float(scale),
                                          out.data_ptr(), _stream()), 'axt_preprocess_u16')
    return out


def tile_occupancy_bytes(frames):
    """Per-tile occupancy of this block of frames as a device u8 tensor [tile_rows * tile_cols] (1 = some pixel of
    the tile is non-zero at some time point)."""
    T_all, H, W = frames.shape
    nty, ntx = -(-H // TILE), -(-W // TILE)
    occ = torch.empty(nty * ntx, dtype=torch.uint8, device=frames.device)
    lib = _lib.load()
    with torch.cuda.device(frames.device):
        _lib.check(lib.axt_tile_occupancy(frames.data_ptr(), T_all, H, W, occ.data_ptr(), _stream()),
                   'axt_tile_occupancy')
    return occ


def tile_list(occ, H, W):
    """Occupancy bytes -> row-major list of kept (tile_row, tile_col)."""
    nty, ntx = -(-H // TILE), -(-W // TILE)
    occ = occ.cpu().numpy().reshape(nty, ntx)
    return [tuple(int(v) for v in ix) for ix in np.argwhere(occ)]


def tile_occupancy(frames):
    """Kept tiles, row-major list of (tile_row, tile_col) (Timelapse.py:551-558)."""
    return tile_list(tile_occupancy_bytes(frames), frames.shape[1], frames.shape[2])


def decode_stitch_nms(yolo, tile_yx, conf_thr=CONF_FLOOR, min_dist=23, cap=None):
    """_yolo_Y2pandas_det + stitch_tiles + _non_max_supression per frame
    (AxonDetections.py:122-128). yolo [n_frames, n_tiles, 12,12,3] on the GPU.
    Returns device tensors conf f32 [F,cap], x i32, y i32, count i32 [F]."""
    n_frames, n_tiles = yolo.shape[0], yolo.shape[1]
    tile_yx = np.ascontiguousarray(tile_yx, np.int32).reshape(-1, 2)
    assert len(tile_yx) == n_tiles and yolo.is_contiguous()
    cap = cap or n_tiles * CELLS
    dev = yolo.device
    # one zeroed block for the four outputs (one fill launch instead of four; slots beyond a frame's count stay zero)
    slots = n_frames * cap
    sec = (slots + 63) // 64 * 64                        # every output starts 256-byte aligned, like a tensor of its own
    block = torch.zeros((3 * sec + n_frames,), dtype=torch.int32, device=dev)
    conf = block[:slots].view(torch.float32).view(n_frames, cap)
    x = block[sec:sec + slots].view(n_frames, cap)
    y = block[2 * sec:2 * sec + slots].view(n_frames, cap)
    count = block[3 * sec:]
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.axt_decode_stitch_nms(yolo.data_ptr(), n_frames, n_tiles, tile_yx.ctypes.data,
                                             ctypes.c_float(conf_thr), int(min_dist), cap, conf.data_ptr(),
                                             x.data_ptr(), y.data_ptr(), count.data_ptr(), _stream()),
                   'axt_decode_stitch_nms')
    return conf, x, y, count


def obs_costs(conf, count, method='scale_to_max', max_conf_cost=4.6):
    """conf capping + observation_model (AxonDetections.py:655-659, mincostflow_models.py:6-27)."""
    n_frames, cap = conf.shape
    cost = torch.zeros((n_frames, cap), dtype=torch.float64, device=conf.device)
    m = {'scale_to_max': 0, 'ceil': 1}[method]
    lib = _lib.load()
    with torch.cuda.device(conf.device):
        _lib.check(lib.axt_obs_costs(conf.data_ptr(), count.data_ptr(), n_frames, cap, m, float(max_conf_cost),
                                     cost.data_ptr(), _stream()), 'axt_obs_costs')
    return cost


class Grid:
    """Masked grid handle (axt_grid): byte mask, bit rows and connected-component labels on the GPU."""

    def __init__(self, mask, conn8=False, device='cuda:0'):
        _require_gpu()
        m = np.ascontiguousarray(np.asarray(mask) == 1, np.uint8)
        self.H, self.W, self.conn8 = int(m.shape[0]), int(m.shape[1]), bool(conn8)
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        with torch.cuda.device(torch.device(device)):
            _lib.check(self._lib.axt_grid_create(m.ctypes.data, self.H, self.W, int(self.conn8), ctypes.byref(h)),
                       'axt_grid_create')
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._lib.axt_grid_destroy(h)


def path_cost(xa, ya, xb, yb, H, W, mask=None, max_dist=MAX_PX_ASSOC_DIST, conn8=False):
    """A* path-length matrix of one frame pair (AxonDetections.py:526-629,717-752)."""
    na, nb = xa.numel(), xb.numel()
    D = torch.empty((na, nb), dtype=torch.int32, device=xa.device)
    lib = _lib.load()
    if mask is not None and not isinstance(mask, Grid):
        mask = Grid(mask.cpu().numpy() if isinstance(mask, torch.Tensor) else mask, conn8, xa.device)
    with torch.cuda.device(xa.device):
        _lib.check(lib.axt_path_cost(xa.data_ptr(), ya.data_ptr(), na, xb.data_ptr(), yb.data_ptr(), nb,
                                     mask._h if mask is not None else None, H, W, int(max_dist), int(bool(conn8)),
                                     D.data_ptr(), _stream()), 'axt_path_cost')
    return D


def path_cells(xa, ya, xb, yb, H, W, mask, max_dist=MAX_PX_ASSOC_DIST, conn8=False):
    """The A* paths of one frame pair on a masked grid (utils.py:379-387): (D i32 [na,nb], cells i32 [na,nb,max_dist]),
    cells[i,j,:D[i,j]] = y*W + x along one minimum-cost path, source first, for the pairs with D < max_dist."""
    na, nb = xa.numel(), xb.numel()
    D = torch.empty((na, nb), dtype=torch.int32, device=xa.device)
    cells = torch.full((na, nb, int(max_dist)), -1, dtype=torch.int32, device=xa.device)
    if not isinstance(mask, Grid):
        mask = Grid(mask.cpu().numpy() if isinstance(mask, torch.Tensor) else mask, conn8, xa.device)
    with torch.cuda.device(xa.device):
        _lib.check(_lib.load().axt_path_cells(xa.data_ptr(), ya.data_ptr(), na, xb.data_ptr(), yb.data_ptr(), nb, mask._h,
                                              H, W, int(max_dist), int(bool(conn8)), D.data_ptr(), cells.data_ptr(),
                                              _stream()), 'axt_path_cells')
    return D, cells


def box_histograms(frames, x, y, count, t_offset=2, box=AXON_BOX_SIZE):
    """feature_model (mincostflow_models.py:30-65) for every detection: (hist f32 [F,cap,180], bin sums f64 [F,cap]).
    frames f32 [T_all,H,W] on the GPU; detection frame f is shown frame f + t_offset (its centre frame)."""
    n_frames, cap = x.shape
    T_all, H, W = frames.shape
    hist = torch.zeros((n_frames, cap, 180), dtype=torch.float32, device=x.device)
    hsum = torch.zeros((n_frames, cap), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().axt_box_histograms(frames.data_ptr(), T_all, H, W, int(t_offset), x.data_ptr(), y.data_ptr(),
                                                  count.data_ptr(), n_frames, cap, int(box), hist.data_ptr(),
                                                  hsum.data_ptr(), _stream()), 'axt_box_histograms')
    return hist, hsum


def build_arcs(x, y, count, H, W, dmax, cost_units=None, mask=None, max_dist=MAX_PX_ASSOC_DIST, conn8=False, vis=None,
               length_table=None, src_count=None):
    """Admissible transition arcs of the whole timelapse, CSR by tail detection (global numbering).
    cost_units: optional int64 [max_gap, max_dist+1] = round(transition cost * 1e6) per (gap, D).
    vis: None, or dict(hist, hsum, weight, miss_rate, thr) -- the appearance term (MCF_VIS_SIM_WEIGHT > 0): costs
    are then computed per pair on the GPU and cost_units is ignored.
    length_table: None, or i16 device tensor [F, cap, max_gap, cap] of precomputed path lengths (<= 0: none), e.g.
    from a path cache written by the reference; mask is then ignored.
    src_count: None, or i32 [F]: build rows only for the first src_count[t] detections of frame t (a frame-sharded
    rank: its own frames' counts, zeros elsewhere); numbering and costs stay those of the whole timelapse.
    Returns device tensors (row_ptr i64 [n_frames*cap+1], col i32, length i16, gap u8, cost i64|None)."""
    n_frames, cap = x.shape
    max_gap = len(dmax)
    dev = x.device
    h_dmax = np.ascontiguousarray(dmax, np.int32)
    row_ptr = torch.empty((n_frames * cap + 1,), dtype=torch.int64, device=dev)
    n_work = n_frames * cap * max_gap + n_frames + 1 + max_gap + 4
    if length_table is not None:
        assert tuple(length_table.shape) == (n_frames, cap, max_gap, cap) and length_table.dtype == torch.int16
        mask = None
    if mask is not None:
        if not isinstance(mask, Grid):
            mask = Grid(mask.cpu().numpy() if isinstance(mask, torch.Tensor) else mask, conn8, dev)
        n_work += (n_frames * cap * max_gap * cap + 1) // 2
    work = torch.empty((n_work,), dtype=torch.int32, device=dev)
    n_arcs = ctypes.c_int64(0)
    lib = _lib.load()
    head = (x.data_ptr(), y.data_ptr(), count.data_ptr(), n_frames, cap, mask._h if mask is not None else None, H, W, int(max_dist),
            int(bool(conn8)), max_gap, h_dmax.ctypes.data)
    if vis is not None:
        head += (vis['hist'].data_ptr(), vis['hsum'].data_ptr(), float(vis['weight']), float(vis['miss_rate']), float(vis['thr']))
    head += (row_ptr.data_ptr(), work.data_ptr())

    def call(col, length, gap, cu, cost, what):
        if src_count is not None or (length_table is not None and vis is not None):
            v = vis or {}
            rc = lib.axt_build_arcs_rows(x.data_ptr(), y.data_ptr(), count.data_ptr(), _lib.dptr(src_count), n_frames, cap,
                                         mask._h if mask is not None else None, H, W, int(max_dist), int(bool(conn8)), max_gap,
                                         h_dmax.ctypes.data, v['hist'].data_ptr() if vis else None,
                                         v['hsum'].data_ptr() if vis else None, float(v.get('weight', 0.0)),
                                         float(v.get('miss_rate', 0.0)), float(v.get('thr', 0.0)), row_ptr.data_ptr(),
                                         work.data_ptr(), _lib.dptr(col), _lib.dptr(length), _lib.dptr(gap), _lib.dptr(cu),
                                         _lib.dptr(cost), ctypes.byref(n_arcs), _lib.dptr(length_table), _stream())
        elif length_table is not None:
            rc = lib.axt_build_arcs_from_lengths(length_table.data_ptr(), x.data_ptr(), y.data_ptr(), count.data_ptr(), n_frames,
                                                 cap, int(max_dist), max_gap, h_dmax.ctypes.data, row_ptr.data_ptr(),
                                                 work.data_ptr(), _lib.dptr(col), _lib.dptr(length), _lib.dptr(gap),
                                                 _lib.dptr(cu), _lib.dptr(cost), ctypes.byref(n_arcs), _stream())
        elif vis is None:
            rc = lib.axt_build_arcs(*head, _lib.dptr(col), _lib.dptr(length), _lib.dptr(gap), _lib.dptr(cu), _lib.dptr(cost),
                                    ctypes.byref(n_arcs), _stream())
        else:
            rc = lib.axt_build_arcs_vis(*head, _lib.dptr(col), _lib.dptr(length), _lib.dptr(gap), _lib.dptr(cost),
                                        ctypes.byref(n_arcs), _stream())
        _lib.check(rc, f'axt_build_arcs({what})')

    with torch.cuda.device(dev):
        call(None, None, None, None, None, 'count')
        n = max(int(n_arcs.value), 1)
        col = torch.empty((n,), dtype=torch.int32, device=dev)
        length = torch.empty((n,), dtype=torch.int16, device=dev)
        gap = torch.empty((n,), dtype=torch.uint8, device=dev)
        cost = cu = None
        if vis is not None:
            cost = torch.empty((n,), dtype=torch.int64, device=dev)
        elif cost_units is not None:
            cu = cost_units if isinstance(cost_units, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(cost_units, np.int64)).to(dev)
            assert cu.shape == (max_gap, max_dist + 1)
            cost = torch.empty((n,), dtype=torch.int64, device=dev)
        call(col, length, gap, cu, cost, 'fill')
    n = int(n_arcs.value)
    return row_ptr, col[:n], length[:n], gap[:n], (cost[:n] if cost is not None else None)


HUNGARIAN_NO_LINK = 0x3fffffffffffffff          # entry of a cost table where a link is not admitted


def hungarian_assoc(x, y, count, H, W, dmax, cost_units, thr_units, max_dist=MAX_PX_ASSOC_DIST, conn8=False,
                    frame_range=None, group=None, mask=None, ctab=None):
    """Frame-to-frame Hungarian association (BASELINE config 3) of a whole timelapse on the GPU.
    frame_range=(a, b): this rank solves only the pairs of source frames a..b-1; the link arrays are then
    combined over `group` with one MAX all-reduce and every rank numbers the chains (frame-sharded runs).
    mask: None (all-ones) or a Grid: path lengths then come from the masked-grid searches of the arc builder.
    ctab: optional i64 device tensor [F, cap, len(dmax), cap] of link costs (HUNGARIAN_NO_LINK where not admitted), used
    instead of the costs derived from the path lengths (axt_hungarian_pairs_costs; the appearance term).
    Returns (track i32 [F,cap] device tensor, n_tracks device tensor [1])."""
    n_frames, cap = x.shape
    max_gap = len(dmax)
    dev = x.device
    h_dmax = np.ascontiguousarray(dmax, np.int32)
    cu = cost_units if isinstance(cost_units, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(cost_units, np.int64)).to(dev)
    slots = n_frames * cap
    pred = torch.empty((2 * slots,), dtype=torch.int32, device=dev)
    work = torch.empty((2 * slots + n_frames + 1,), dtype=torch.int32, device=dev)
    track = torch.empty((n_frames, cap), dtype=torch.int32, device=dev)
    n_tracks = torch.empty((1,), dtype=torch.int32, device=dev)       # (axt_chain_tracks always writes it)
    a, b = (0, n_frames) if frame_range is None else frame_range
    lib = _lib.load()
    with torch.cuda.device(dev):
        if ctab is not None:
            if ctab.dtype != torch.int64 or tuple(ctab.shape) != (n_frames, cap, max_gap, cap) or not ctab.is_contiguous():
                raise ValueError(f'ctab must be a contiguous int64 tensor [{n_frames}, {cap}, {max_gap}, {cap}]')
            _lib.check(lib.axt_hungarian_pairs_costs(x.data_ptr(), y.data_ptr(), count.data_ptr(), n_frames, cap, max_gap,
                                                     ctab.data_ptr(), int(thr_units), int(a), int(b), pred.data_ptr(),
                                                     work.data_ptr(), _stream()), 'axt_hungarian_pairs_costs')
        else:
            _lib.check(lib.axt_hungarian_pairs_grid(x.data_ptr(), y.data_ptr(), count.data_ptr(), n_frames, cap,
                                                    mask._h if mask is not None else None, H, W,
                                                    int(max_dist), int(bool(conn8)), max_gap, h_dmax.ctypes.data,
                                                    cu.data_ptr(), int(thr_units), int(a), int(b), pred.data_ptr(),
                                                    work.data_ptr(), _stream()), 'axt_hungarian_pairs_grid')
        if frame_range is not None:
            import torch.distributed as dist
            from . import sharded
            sharded._collective('hungarian_links_allreduce', lambda: dist.all_reduce(pred, op=dist.ReduceOp.MAX, group=group))
        _lib.check(lib.axt_chain_tracks(count.data_ptr(), n_frames, cap, pred.data_ptr(), work.data_ptr(),
                                        track.data_ptr(), n_tracks.data_ptr(), _stream()), 'axt_chain_tracks')
    return track, n_tracks


def ided_table(track, conf, x, y, count, n_ids, label_quirk=True, id_row=None, n_rows=None):
    """IDed_dets_all's values (AxonDetections.py:825-842) as a pinned host f64 array [n_rows, 3*F]: filled on the
    GPU (axt_ided_table), copied across once. id_row: optional i32 device tensor id -> row for ids with gaps.
    Returns (array, wait): the copy is asynchronous, wait() returns when the array holds the table."""
    n_frames, cap = x.shape
    dev = x.device
    n_rows = int(n_ids if n_rows is None else n_rows)
    table = torch.empty((n_rows, 3 * n_frames), dtype=torch.float64, device=dev)
    work = torch.empty((n_frames,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().axt_ided_table(track.data_ptr(), conf.data_ptr(), x.data_ptr(), y.data_ptr(),
                                              count.data_ptr(), n_frames, cap, int(n_ids),
                                              id_row.data_ptr() if id_row is not None else None, n_rows,
                                              int(bool(label_quirk)), work.data_ptr(), table.data_ptr(), _stream()),
                   'axt_ided_table')
    host = torch.empty(table.shape, dtype=torch.float64, pin_memory=True)
    host.copy_(table, non_blocking=True)             # the caller wraps the array while the copy is in flight ...
    return host.numpy(), torch.cuda.current_stream(dev).synchronize       # ... and calls this before handing it on


def detection_confusion(conf, x, y, count, gx, gy, gcount, thrs, min_dist=23, k_mask=-1):
    """compute_TP_FP_FN (AxonDetections.py:409-466) for all frames and thresholds at once: i32 [F,3,n_thr] on the
    device (TP, FP, FN); with k_mask >= 0 also (fp_mask u8 [F,cap], fn_mask u8 [F,gcap]) for that threshold."""
    n_frames, cap = x.shape
    gcap = gx.shape[1]
    dev = x.device
    th = torch.from_numpy(np.array(thrs, np.float64)).to(dev)            # (a copy: the cached thresholds are read-only)
    out = torch.zeros((n_frames, 3, len(th)), dtype=torch.int32, device=dev)
    fp = fn = None
    if k_mask >= 0:
        fp = torch.zeros((n_frames, cap), dtype=torch.uint8, device=dev)
        fn = torch.zeros((n_frames, gcap), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().axt_detection_confusion(conf.data_ptr(), x.data_ptr(), y.data_ptr(), count.data_ptr(), n_frames,
                                                       cap, gx.data_ptr(), gy.data_ptr(), gcount.data_ptr(), gcap,
                                                       th.data_ptr(), len(th), int(min_dist), int(k_mask), out.data_ptr(),
                                                       _lib.dptr(fp), _lib.dptr(fn), _stream()), 'axt_detection_confusion')
    return (out, fp, fn) if k_mask >= 0 else out


_HOST_STAGING = {}


def to_host(*tensors):
    """Device tensors -> numpy arrays through pinned staging buffers (kept per dtype and grown as needed): the arc list of a
    300 k-detection timelapse is 120 MB, which a pageable copy moves at a few GB/s and a pinned one at PCIe speed. The copies
    are enqueued together and waited for once. The arrays are views of the staging buffers: valid until the next call."""
    out = []
    for k, t in enumerate(tensors):
        n = int(t.numel())
        key = (k, t.dtype)
        buf = _HOST_STAGING.get(key)
        if buf is None or buf.numel() < n:
            buf = _HOST_STAGING[key] = torch.empty((max(n, 1) * 5 // 4 + 16,), dtype=t.dtype, pin_memory=True)
        buf[:n].copy_(t.reshape(-1), non_blocking=True)
        out.append(buf[:n].numpy().reshape(tuple(t.shape)))
    # a non_blocking D2H copy runs on the current stream of the SOURCE tensor's device, which need not be the current
    # device (Detector(device='cuda:1') without set_device): wait on every device that holds one of the tensors
    for dev in {t.device for t in tensors if t.is_cuda}:
        torch.cuda.current_stream(dev).synchronize()
    return out


def arc_cost_int(cost, kind, a, b):
    return int(_lib.load().axt_arc_cost_int(float(cost), int(kind), int(a), int(b)))


def mcf_solve(obs_int, entry_int, exit_int, row_ptr, col, cost_int, min_flow, max_flow, duals=False):
    """MinCostFlowTracker.compute_trajectories (AxonDetections.py:690) on host arrays.
    Returns (next i32 [n], track i32 [n], n_tracks, total_cost) or None when infeasible; with duals=True a fifth element
    (pot_u i64 [n], pot_v i64 [n], pot_t): the node potentials that certify the optimum (axt_mcf_solve_duals)."""
    n = len(obs_int)
    arrs = [np.ascontiguousarray(a, np.int64) for a in (obs_int, entry_int, exit_int, row_ptr)]
    col = np.ascontiguousarray(col, np.int32)
    cost_int = np.ascontiguousarray(cost_int, np.int64)
    nxt = np.empty(n, np.int32)
    track = np.empty(n, np.int32)
    n_tracks, total = ctypes.c_int(0), ctypes.c_int64(0)
    lib = _lib.load()
    if duals:
        pu, pv, pt = np.zeros(n, np.int64), np.zeros(n, np.int64), ctypes.c_int64(0)
        rc = _lib.check(lib.axt_mcf_solve_duals(n, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                                arrs[3].ctypes.data, col.ctypes.data, cost_int.ctypes.data, int(min_flow),
                                                int(max_flow), nxt.ctypes.data, track.ctypes.data, ctypes.byref(n_tracks),
                                                ctypes.byref(total), pu.ctypes.data, pv.ctypes.data, ctypes.byref(pt)),
                        'axt_mcf_solve_duals')
        return None if rc == 1 else (nxt, track, int(n_tracks.value), int(total.value), (pu, pv, int(pt.value)))
    rc = _lib.check(lib.axt_mcf_solve(n, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                      arrs[3].ctypes.data, col.ctypes.data, cost_int.ctypes.data, int(min_flow),
                                      int(max_flow), nxt.ctypes.data, track.ctypes.data, ctypes.byref(n_tracks),
                                      ctypes.byref(total)), 'axt_mcf_solve')
    if rc == 1:
        return None
    return nxt, track, int(n_tracks.value), int(total.value)


class McfShard:
    """One rank's part of the frame-sharded flow solve (axt_mcf_shard_*): begin() solves this rank's run of time blocks
    and returns its state (uint8 array) for the all-gather; finish(states) joins the runs and returns what mcf_solve
    returns. The network arrays are kept alive by this object."""

    def __init__(self, obs_int, entry_int, exit_int, row_ptr, col, cost_int, rank, world):
        self._lib = _lib.load()
        self.n = len(obs_int)
        self._keep = [np.ascontiguousarray(a, np.int64) for a in (obs_int, entry_int, exit_int, row_ptr)]
        self._keep += [np.ascontiguousarray(col, np.int32), np.ascontiguousarray(cost_int, np.int64)]
        h, nbytes = ctypes.c_void_p(), ctypes.c_int64(0)
        k = self._keep
        _lib.check(self._lib.axt_mcf_shard_begin(self.n, k[0].ctypes.data, k[1].ctypes.data, k[2].ctypes.data, k[3].ctypes.data,
                                                 k[4].ctypes.data, k[5].ctypes.data, int(rank), int(world), ctypes.byref(h),
                                                 ctypes.byref(nbytes)), 'axt_mcf_shard_begin')
        self._h, self.rank, self.world = h, int(rank), int(world)
        self.state = np.zeros(int(nbytes.value), np.uint8)
        _lib.check(self._lib.axt_mcf_shard_export(self._h, self.state.ctypes.data), 'axt_mcf_shard_export')

    def finish(self, states, min_flow, max_flow, duals=False):
        """states: list over ranks of uint8 arrays (this rank's own entry is not read). duals: as mcf_solve."""
        states = [np.ascontiguousarray(s, np.uint8) for s in states]
        ptrs = (ctypes.c_void_p * self.world)(*[s.ctypes.data if len(s) else None for s in states])
        sizes = np.array([len(s) for s in states], np.int64)
        nxt, track = np.empty(self.n, np.int32), np.empty(self.n, np.int32)
        n_tracks, total = ctypes.c_int(0), ctypes.c_int64(0)
        if duals:
            pu, pv, pt = np.zeros(self.n, np.int64), np.zeros(self.n, np.int64), ctypes.c_int64(0)
            rc = _lib.check(self._lib.axt_mcf_shard_finish_duals(self._h, ptrs, sizes.ctypes.data, int(min_flow), int(max_flow),
                                                                 nxt.ctypes.data, track.ctypes.data, ctypes.byref(n_tracks),
                                                                 ctypes.byref(total), pu.ctypes.data, pv.ctypes.data,
                                                                 ctypes.byref(pt)), 'axt_mcf_shard_finish_duals')
            return None if rc == 1 else (nxt, track, int(n_tracks.value), int(total.value), (pu, pv, int(pt.value)))
        rc = _lib.check(self._lib.axt_mcf_shard_finish(self._h, ptrs, sizes.ctypes.data, int(min_flow), int(max_flow),
                                                       nxt.ctypes.data, track.ctypes.data, ctypes.byref(n_tracks),
                                                       ctypes.byref(total)), 'axt_mcf_shard_finish')
        return None if rc == 1 else (nxt, track, int(n_tracks.value), int(total.value))

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            self._lib.axt_mcf_shard_free(h)
