"""AxonDetections: the reference's orchestration class (axtrack/AxonDetections.py) re-hosted on
the HIP hot path. Same public surface on the inference path -- detect_dataset(), assign_ids(),
get_frame_dets(), IDed_dets_all, _detections, dir, name, len() -- but a timelapse is processed as
whole-array GPU passes instead of a Python loop over frames and DataFrames:

    detect_dataset : CNN forward for all (frame, kept tile) -> decode+stitch+NMS
    assign_ids     : observation costs + admissible-arc list on the GPU -> min-cost-flow solve (or the
                     frame-to-frame Hungarian variant) -> trajectories -> IDed_dets_all (filled on the GPU)
    compute_TP_FP_FN / get_detection_metrics : the evaluation metrics of labelled data, one launch

pandas objects are only materialised at the API boundary (lazily for the per-frame tables).
"""
import os
import re
import pickle

import numpy as np
import pandas as pd
import torch

from . import hotpath as hp


def transition_cost_table(P, max_px=hp.MAX_PX_ASSOC_DIST, vis_sim=0.0):
    """transition_model (mincostflow_models.py:67-119) tabulated for every integer path length
    D = 0..max_px and gap = 1..MCF_MAX_NUM_MISSES+1, numpy f64 exactly as the reference computes
    it, for one value of the visual similarity. Returns (cost f64 [gaps, max_px+1], dmax i32 [gaps]) with
    dmax = largest D whose cost < MCF_EDGE_COST_THR. With MCF_VIS_SIM_WEIGHT = 0 the table IS the cost model; with
    a weight the costs are computed per pair on the GPU (axt_build_arcs_vis) and the table for vis_sim = 1 only
    bounds the candidate pairs."""
    w = P['MCF_VIS_SIM_WEIGHT']
    gaps = P['MCF_MAX_NUM_MISSES'] + 1
    D = np.arange(0, max_px + 1)
    table = np.empty((gaps, max_px + 1))
    dmax = np.zeros(gaps, np.int32)
    for g in range(1, gaps + 1):
        distances = ((D / max_px) - 1) * -1
        with np.errstate(divide='ignore'):
            costs = -np.log((1 - w) * distances * (P['MCF_MISS_RATE'] ** (g - 1)) + w * vis_sim + 1e-6)
        costs[distances == 0] = np.inf
        table[g - 1] = costs
        ok = np.nonzero(costs[1:] < P['MCF_EDGE_COST_THR'])[0]
        dmax[g - 1] = ok.max() + 1 if len(ok) else 0
    return table, dmax


class AxonDetections(object):
    def __init__(self, model, dataset, parameters, directory, timepoint_subset=None):
        self.model = model
        self.dataset = dataset
        self.name = dataset.name
        self.dir = directory
        if self.dir:
            os.makedirs(self.dir, exist_ok=True)
        # AxonDetections.py:52-55: the detection frames to work on (default: all). Detection, association and every
        # per-frame accessor then index POSITIONS in this list, as the reference's do (its loops run over
        # range(len(self)) and look frames up through timepoint_subset).
        if timepoint_subset is None:
            self.timepoint_subset = range(self.dataset.sizet)      # (kept as a range: nothing per frame on the way to the first launch)
        else:
            self.timepoint_subset = [int(t) for t in timepoint_subset]
            if any(t < 0 or t >= self.dataset.sizet for t in self.timepoint_subset):
                raise ValueError(f'timepoint_subset must lie in [0, {self.dataset.sizet})')
        self.P = dict(parameters)
        self.device = dataset.device
        self.Sx, self.Sy, self.tilesize = parameters['SX'], parameters['SY'], parameters['TILESIZE']
        if (self.Sx, self.Sy, self.tilesize) != (hp.S, hp.S, hp.TILE):
            raise ValueError('the HIP detector is specialised for TILESIZE=512, SX=SY=12')
        self.nms_min_dist = parameters.get('NON_MAX_SUPRESSION_DIST')
        self.conf_thr = parameters['BBOX_THRESHOLD']
        self.all_conf_thrs = _conf_thresholds(self.conf_thr)
        self.max_px_assoc_dist = hp.MAX_PX_ASSOC_DIST
        self.axon_box_size = hp.AXON_BOX_SIZE
        self.labelled = False
        self.conn8 = bool(parameters.get('ASTAR_8_CONNECTED', False))
        self.reproduce_label_quirk = bool(parameters.get('REPRODUCE_FRAME_LABEL_QUIRK', True))
        self._det_tables = None
        self._n_ids, self._n_tracks_dev = None, None

    @property
    def n_ids(self):
        """Number of identities of the last association (None: adopted from a cache, ids are whatever the cache holds). The
        frame-to-frame variant leaves the count on the device; it is fetched when somebody asks."""
        if self._n_ids is None and self._n_tracks_dev is not None:
            self._n_ids = int(self._n_tracks_dev.item())
        return self._n_ids

    @n_ids.setter
    def n_ids(self, value):
        self._n_ids, self._n_tracks_dev = value, None

    def __len__(self):
        """Number of detection frames (after a multi-GPU gather: of the whole timelapse)."""
        d_count = getattr(self, 'd_count', None)
        return int(d_count.shape[0]) if d_count is not None else len(self.timepoint_subset)

    # ------------------------------------------------------------------ caches (AxonDetections.py:141-176)
    def _cache_fname(self, which):
        return f'{self.dir}/{self.dataset.name}_{which}.pkl'

    def from_cache(self, which):
        with open(self._cache_fname(which), 'rb') as file:
            return pickle.load(file)

    def to_cache(self, which, dat):
        with open(self._cache_fname(which), 'wb') as file:
            pickle.dump(dat, file)

    # ------------------------------------------------------------------ detection (AxonDetections.py:87-139)
    def detect_dataset(self, cache=None):
        if cache == 'from':
            self._set_detections_from_tables(self.from_cache('_detections'))
            return
        frames = self.dataset.frames
        if hasattr(self.model, 'set_arith'):
            self.model.set_arith(self.P.get('CNN_ARITH', 'f32'))       # read at inference time, like every parameter
        streamed = self._detect_streaming() if getattr(self.dataset, '_pending', False) else None
        self.tile_yx = self.dataset.tile_yx
        if not self.tile_yx:
            raise ValueError('the timelapse is empty (no tile has a non-zero pixel)')
        # the frames of timepoint_subset (AxonDetections.py:111), one launch sequence per run of consecutive frames
        sub = self.timepoint_subset
        whole = isinstance(sub, range)
        runs, a = [], 0
        if whole:
            runs = [(0, len(sub))]
        else:
            for k in range(1, len(sub) + 1):
                if k == len(sub) or sub[k] != sub[k - 1] + 1:
                    runs.append((sub[a], k - a))
                    a = k
        if streamed is not None and streamed[0] == self.tile_yx and whole:
            self._yolo = streamed[1]                                 # computed chunk by chunk beside the copies
        else:
            parts = [self.model.detect_frames(frames, self.tile_yx, t0, n) for t0, n in runs]
            self._yolo = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        self._tiled_tables = None
        thr = float(np.float32(self.all_conf_thrs.min()))
        self.d_conf, self.d_x, self.d_y, self.d_count = hp.decode_stitch_nms(
            self._yolo, self.tile_yx, thr, self.nms_min_dist)
        import torch.distributed as dist
        if self.d_conf.shape[1] > 2048 and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            # large frames (more than 14 tiles): the arrays were sized for one detection per grid cell; keep what the fullest
            # frame needs (the arc builder's scratch and the frame-to-frame variant scale with the capacity). Frame-sharded
            # ranks agree on a common capacity in gather_detections().
            self._shrink_capacity(int(self.d_count.max().item()))
        self._det_tables = None
        self._host = None
        if cache == 'to':
            self.to_cache('_detections', self._detections)

    def _detect_streaming(self):
        """Host-resident input (Timelapse.from_host_u16): the CNN of the detection frames every chunk completes is
        enqueued right behind the chunk's preprocessing, while the next chunk's H2D copy runs on the copy stream. The
        kept-tile list is a property of the whole timelapse (Timelapse.py:551-558) and only known after the last chunk:
        the chunks are detected with every tile, which is the final list unless some tile is empty at EVERY time point
        (then detect_dataset repeats the detection on the resident frames with the right list). Returns (tile list used,
        YOLO grids [sizet, n_tiles, 12,12,3])."""
        ds = self.dataset
        tiles = [(ty, tx) for ty in range(ds.ytiles) for tx in range(ds.xtiles)]
        ctx = ds.temporal_context
        # the front of the network (conv blocks 0-5: 87 % of the FLOPs) chunk by chunk, the rest once for all frames: the first
        # linear layer streams its 168 MB of weights per call, whatever the batch
        split = ds.sizet * len(tiles) <= getattr(self.model, 'max_batch', 0)
        parts, done, occ = [], 0, None
        for a, b, occ in ds.stream_chunks():
            ready = min(max(b - 2 * ctx, 0), ds.sizet)               # detection frame t reads input frames t .. t + 2 ctx
            if ready > done:
                if split:
                    self.model.front_frames(ds.frames, tiles, done, ready - done, done * len(tiles))
                else:
                    parts.append(self.model.detect_frames(ds.frames, tiles, done, ready - done))
                done = ready
        yolo = self.model.back(ds.sizet, len(tiles)) if split else (parts[0] if len(parts) == 1 else torch.cat(parts, 0))
        ds._occ_done.synchronize()                                    # (a few bytes, behind the last chunk's preprocessing)
        ds._tile_yx = hp.tile_list(ds._occ_host, ds.sizey, ds.sizex)
        return tiles, yolo

    def _shrink_capacity(self, fullest):
        cap = min(-(-max(int(fullest), 1) // 64) * 64, int(self.d_conf.shape[1]))
        self.d_conf, self.d_x, self.d_y = (a[:, :cap].contiguous() for a in (self.d_conf, self.d_x, self.d_y))

    def gather_detections(self, group=None):
        """Frame-sharded runs: every rank has detected its own contiguous block of frames; one
        all-gather (RCCL over xGMI on GPUs) gives every rank the detections of the whole
        timelapse, in rank order, before the global flow solve. Blocks must have equal length."""
        from .sharded import all_gather_detections
        import torch.distributed as dist
        local = int(self.d_count.shape[0])
        if self.d_conf.shape[1] > 2048 and dist.is_initialized() and dist.get_world_size(group) > 1:
            # large frames: every rank keeps what the fullest frame of the WHOLE timelapse needs (one MAX all-reduce over
            # the group of the gather), so that the gathered arrays fit the association kernels' capacity
            from .sharded import _collective
            fullest = self.d_count.max().reshape(1).clone()
            _collective('capacity_allreduce', lambda: dist.all_reduce(fullest, op=dist.ReduceOp.MAX, group=group))
            self._shrink_capacity(int(fullest.item()))
        if self.P['MCF_VIS_SIM_WEIGHT'] and dist.is_initialized() and dist.get_world_size(group) > 1:
            # the appearance features need the pixels, which only the owning rank has: they travel with the detections
            hist, hsum = self._appearance()
            world = dist.get_world_size(group)
            g_hist = torch.empty((world * local,) + tuple(hist.shape[1:]), dtype=hist.dtype, device=hist.device)
            g_hsum = torch.empty((world * local, hsum.shape[1]), dtype=hsum.dtype, device=hsum.device)
            dist.all_gather_into_tensor(g_hist, hist.contiguous(), group=group)
            dist.all_gather_into_tensor(g_hsum, hsum.contiguous(), group=group)
            self._hist = (g_hist, g_hsum)
        self.d_conf, self.d_x, self.d_y, self.d_count = all_gather_detections(
            self.d_conf, self.d_x, self.d_y, self.d_count, group,
            check_shapes=not getattr(self.dataset, '_gather_shapes_agree', False))
        self.dataset._gather_shapes_agree = True          # the shapes follow from the timelapse: checked once
        self._host, self._det_tables = None, None
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            r = dist.get_rank(group)
            self._shard = (r * local, (r + 1) * local, group)       # this rank's frames within the gathered arrays

    def set_detections(self, conf, x, y, count):
        """Adopt detection lists that are already arrays (conf f32 [F,cap], x / y i32 [F,cap], count i32 [F], every frame
        in descending confidence): association-only workloads (bench.py --workload assoc-*) and detections produced
        elsewhere. The arrays go to the dataset's device; detect_dataset() is not needed afterwards."""
        dev = self.device
        as_dev = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(device=dev, dtype=dt).contiguous()
        self.d_conf, self.d_x, self.d_y, self.d_count = (as_dev(conf, torch.float32), as_dev(x, torch.int32), as_dev(y, torch.int32),
                                                         as_dev(count, torch.int32))
        if self.d_conf.dim() != 2 or self.d_x.shape != self.d_conf.shape or self.d_y.shape != self.d_conf.shape \
                or self.d_count.shape != (self.d_conf.shape[0],):
            raise ValueError('conf, x, y must be [F, cap] and count [F]')
        self._host, self._det_tables, self._yolo, self._tiled_tables = None, None, None, None

    def _host_dets(self):
        """(count i32 [F], conf f32 [F,cap], x, y) on the host, fetched once."""
        if getattr(self, '_host', None) is None:
            self._host = (self.d_count.cpu().numpy(), self.d_conf.cpu().numpy(), self.d_x.cpu().numpy(),
                          self.d_y.cpu().numpy())
        return self._host

    def _set_detections_from_tables(self, tables):
        F = len(tables)
        cap = max([len(t) for t in tables] + [1])
        conf = np.zeros((F, cap), np.float32); x = np.zeros((F, cap), np.int32); y = np.zeros((F, cap), np.int32)
        cnt = np.zeros(F, np.int32)
        for f, t in enumerate(tables):
            n = len(t)
            cnt[f] = n
            conf[f, :n] = t.conf.to_numpy(dtype=np.float32)
            x[f, :n] = t.anchor_x.to_numpy(dtype=np.int32)
            y[f, :n] = t.anchor_y.to_numpy(dtype=np.int32)
        dev = self.device
        self.d_conf, self.d_x, self.d_y, self.d_count = (torch.from_numpy(a).to(dev) for a in (conf, x, y, cnt))
        self._host, self._det_tables = (cnt, conf, x, y), list(tables)

    @property
    def _detections(self):
        """list of per-frame DataFrames [conf Float32, anchor_x Int64, anchor_y Int64], index
        Axon_000.. in descending confidence (AxonDetections.py:235,261,276-277)."""
        if self._det_tables is None:
            cnt, conf, x, y = self._host_dets()
            tabs = []
            for f in range(len(cnt)):
                n = int(cnt[f])
                tabs.append(pd.DataFrame({'conf': pd.array(conf[f, :n], dtype='Float32'),
                                          'anchor_x': pd.array(x[f, :n].astype(np.int64), dtype='Int64'),
                                          'anchor_y': pd.array(y[f, :n].astype(np.int64), dtype='Int64')},
                                         index=[f'Axon_{i:0>3}' for i in range(n)]))
            self._det_tables = tabs
        return self._det_tables

    @property
    def _pandas_tiled_dets(self):
        """_yolo_Y2pandas_det (AxonDetections.py:178-248) of every frame: per frame a list over the kept tiles of tables
        [conf Float32, anchor_x Int64, anchor_y Int64] in TILE coordinates, the cells at or above the confidence floor
        named Axon_000.. in cell order and sorted by ascending confidence -- the tables before stitching and NMS. The
        hot path never needs them (its kernel goes from the YOLO grids to the frame's final list), so they are decoded
        from the grids on the host when first asked for: f32 arithmetic, half-to-even rounding, all-zero cells left
        at zero, exactly as :192-210."""
        if getattr(self, '_tiled_tables', None) is None:
            yolo = getattr(self, '_yolo', None)
            if yolo is None:
                raise ValueError('the tile tables come from the YOLO grids: run detect_dataset() (not from a cache)')
            y = yolo.cpu().numpy().astype(np.float32, copy=False)            # [F, n_tiles, Sx, Sy, 3]
            ii = np.arange(self.Sx, dtype=np.float32).reshape(1, 1, self.Sx, 1)
            jj = np.arange(self.Sy, dtype=np.float32).reshape(1, 1, 1, self.Sy)
            ts = np.float32(self.tilesize)
            ax = np.rint(((y[..., 1] + ii) * ts) / np.float32(self.Sx))
            ay = np.rint(((y[..., 2] + jj) * ts) / np.float32(self.Sy))
            zero = (y == 0).all(-1)
            ax[zero] = 0
            ay[zero] = 0
            thr = np.float32(self.all_conf_thrs.min())
            frames = []
            for f in range(y.shape[0]):
                tiles = []
                for k in range(y.shape[1]):
                    conf = y[f, k, ..., 0].reshape(-1)
                    keep = np.nonzero(conf >= thr)[0]
                    order = keep[np.argsort(conf[keep], kind='stable')]       # ties stay in cell order
                    names = np.empty(len(conf), dtype=object)
                    names[keep] = [f'Axon_{i:0>3}' for i in range(len(keep))]
                    tiles.append(pd.DataFrame({'conf': pd.array(conf[order], dtype='Float32'),
                                               'anchor_x': pd.array(ax[f, k].reshape(-1)[order].astype(np.int64), dtype='Int64'),
                                               'anchor_y': pd.array(ay[f, k].reshape(-1)[order].astype(np.int64), dtype='Int64')},
                                              index=list(names[order])))
                frames.append(tiles)
            self._tiled_tables = frames
        return self._tiled_tables

    # ------------------------------------------------------------------ access (AxonDetections.py:280-353)
    def get_frame_dets(self, which_dets, t, libmot=False, unstitched=False):
        if t is None:
            all_dets = [self.get_frame_dets(which_dets, t, libmot) for t in range(len(self))]
            return pd.concat(all_dets, axis=not libmot)
        if unstitched:
            # only for 'all' and 'confident' (AxonDetections.py:322-331): the tile-wise list of tables
            if which_dets == 'all':
                return [d.copy() for d in self._pandas_tiled_dets[t]]
            if which_dets == 'confident':
                return [d[d.conf > self.conf_thr] for d in self._pandas_tiled_dets[t]]
            raise ValueError("unstitched=True goes with which_dets 'all' or 'confident'")
        if which_dets == 'all':
            det = self._detections[t]
        elif which_dets == 'confident':
            det = self._detections[t][self._detections[t].conf > self.conf_thr]
        elif which_dets == 'IDed':
            assert self._IDed_detections, "Run .assign_IDs() first!"
            det = self._IDed_detections[t]
        elif which_dets == 'groundtruth':
            if not self.labelled:
                raise ValueError("no labels: call set_groundtruth() first")
            gx, gy = self._gt[t]
            det = pd.DataFrame({'conf': pd.array(np.ones(len(gx), np.float32), dtype='Float32'),
                                'anchor_x': pd.array(np.asarray(gx, np.int64), dtype='Int64'),
                                'anchor_y': pd.array(np.asarray(gy, np.int64), dtype='Int64')},
                               index=[f'Axon_{i:0>3}' for i in self._gt_ids[t]])
        else:
            raise NotImplementedError(f"which_dets={which_dets!r} is a plotting selection (out of scope)")
        if libmot:
            return self.det2libmot_det(det, t)
        return det.copy()

    def det2libmot_det(self, detection, t):
        """AxonDetections.py:754-784"""
        conf, x, y = detection.values.T
        half = self.axon_box_size // 2
        frame_id = np.full(conf.shape, t)
        boxs = np.full(conf.shape, self.axon_box_size)
        # the reference takes idx[-3:], which folds identities >= 1000 onto 0..999; the whole number is used here
        axon_id = np.array([_axon_number(idx) for idx in detection.index])
        det_libmot = np.stack([frame_id, axon_id, x - half, y - half, boxs, boxs, conf]).T
        cols = ['FrameId', 'Id', 'X', 'Y', 'Width', 'Height', 'conf']
        return pd.DataFrame(det_libmot, columns=cols).set_index(['FrameId', 'Id'])

    # ------------------------------------------------------------------ detection metrics (AxonDetections.py:378-503)
    def set_groundtruth(self, labels):
        """labels: per detection frame (x, y) or (x, y, ids) -- integer anchor arrays and, for the tracking scores of
        search_MCF_params, the axons' identities (the numbers of the labels' Axon_### names; default: position in
        the frame). The reference reads them from the labelled dataset's YOLO targets (get_frame_and_truedets,
        :355-376); the dataset side is out of scope here."""
        if len(labels) != len(self):
            raise ValueError(f'{len(labels)} label frames for {len(self)} detection frames')
        self._gt = [(np.asarray(l[0], np.int64), np.asarray(l[1], np.int64)) for l in labels]
        self._gt_ids = [np.asarray(l[2], np.int64) if len(l) > 2 else np.arange(len(l[0])) for l in labels]
        if any(len(i) != len(x) or len(np.unique(i)) != len(i) for i, (x, _) in zip(self._gt_ids, self._gt)):
            raise ValueError('label identities must be unique within a frame and match the anchors in number')
        gcap = max([len(x) for x, _ in self._gt] + [1])
        gx = np.zeros((len(self), gcap), np.int32); gy = np.zeros((len(self), gcap), np.int32)
        for t, (x, y) in enumerate(self._gt):
            gx[t, :len(x)] = x; gy[t, :len(y)] = y
        dev = self.device
        self._gt_dev = (torch.from_numpy(gx).to(dev), torch.from_numpy(gy).to(dev),
                        torch.tensor([len(x) for x, _ in self._gt], dtype=torch.int32, device=dev))
        self.labelled = True

    def detection_confusion(self):
        """compute_TP_FP_FN('all', t) of every frame in one launch: int array [frames, 3 (TP, FP, FN), 13 thresholds]."""
        if not self.labelled:
            raise ValueError("no labels: call set_groundtruth() first")
        return hp.detection_confusion(self.d_conf, self.d_x, self.d_y, self.d_count, *self._gt_dev, self.all_conf_thrs,
                                      self.nms_min_dist).cpu().numpy().astype(np.int64)

    def compute_TP_FP_FN(self, which_dets, t, return_FP_FN_mask=False):
        """AxonDetections.py:409-466 for one frame and one selection of detections ('all', 'confident', 'IDed')."""
        det = self.get_frame_dets(which_dets, t)
        dev = self.device
        n = len(det)
        conf = torch.zeros((1, max(n, 1)), dtype=torch.float32, device=dev)
        x = torch.zeros((1, max(n, 1)), dtype=torch.int32, device=dev); y = torch.zeros_like(x)
        if n:
            conf[0, :n] = torch.from_numpy(det.conf.to_numpy(dtype=np.float32))
            x[0, :n] = torch.from_numpy(det.anchor_x.to_numpy(dtype=np.int32))
            y[0, :n] = torch.from_numpy(det.anchor_y.to_numpy(dtype=np.int32))
        cnt = torch.tensor([n], dtype=torch.int32, device=dev)
        gx, gy, gc = (v[t:t + 1].contiguous() for v in self._gt_dev)
        k = int(np.where(self.all_conf_thrs == self.conf_thr)[0][0]) if return_FP_FN_mask else -1
        res = hp.detection_confusion(conf, x, y, cnt, gx, gy, gc, self.all_conf_thrs, self.nms_min_dist, k)
        if return_FP_FN_mask:
            _, fp, fn = res
            return fp[0, :n].cpu().numpy().astype(bool), fn[0, :int(gc.item())].cpu().numpy().astype(bool)
        return res[0].cpu().numpy().astype(np.int64)

    def compute_prc_rcl_F1(self, cnfs_mtrx, return_dataframe=False):
        """AxonDetections.py:468-503 (host arithmetic, three lines)."""
        prc = cnfs_mtrx[0] / (cnfs_mtrx[0] + cnfs_mtrx[1] + 1e-6)
        rcl = cnfs_mtrx[0] / (cnfs_mtrx[0] + cnfs_mtrx[2] + 1e-6)
        f1 = 2 * (prc * rcl) / ((prc + rcl) + 1e-6)
        metric = np.array([prc, rcl, f1]).round(3)
        if return_dataframe:
            index = pd.MultiIndex.from_product([('precision', 'recall', 'F1'), self.all_conf_thrs])
            return pd.Series(metric.flatten(), index=index)
        return metric

    def get_detection_metrics(self, which_dets, t, return_all_conf_thrs=False):
        """AxonDetections.py:378-407"""
        if not self.labelled:
            return None, None, None
        prc_rcl_f1 = self.compute_prc_rcl_F1(self.compute_TP_FP_FN(which_dets, t))
        if not return_all_conf_thrs:
            return prc_rcl_f1[:, np.where(self.all_conf_thrs == self.conf_thr)[0][0]]
        return prc_rcl_f1

    # ------------------------------------------------------------------ association (AxonDetections.py:505-524)
    def assign_ids(self, astar_paths_cache=None, assigedIDs_cache=None, _len_table=None):
        """AxonDetections.py:505-524. astar_paths_cache: 'from' adopts the path lengths of a
        '{name}_astar_dets_paths.pkl' (the reference's format: per frame pair a nested list of coo matrices / None)
        instead of computing them; 'to' writes such a file (astar_dets_paths)."""
        self._len_table = _len_table
        if assigedIDs_cache != 'from':
            if astar_paths_cache == 'from':
                self._len_table = self._length_table_from_paths(self.from_cache('astar_dets_paths'))
            elif astar_paths_cache == 'to':
                self.to_cache('astar_dets_paths', self.astar_dets_paths())
        if assigedIDs_cache == 'from':
            self._set_ided_from_tables(self.from_cache('_IDed_detections'))
        else:
            self._solved = self._assign_IDs_to_detections()
            self._ided_tables = None
            if assigedIDs_cache == 'to' and self._solved:
                self.to_cache('_IDed_detections', self._IDed_detections)
        self.IDed_dets_all = self._agg_all_IDed_dets() if self._solved else None

    @property
    def _IDed_detections(self):
        """list of per-frame DataFrames of the IDed detections, rows sorted by ID, index Axon_{id:03}
        (libmot_det2det, AxonDetections.py:786-823); None if the flow problem was infeasible (:691-696).
        Built on first access -- the hot path itself only keeps arrays."""
        if not getattr(self, '_solved', False):
            return None
        if self._ided_tables is None:
            cnt, conf, x, y = self._host_dets()
            frame, tid, c, xx, yy = self.ided_arrays()
            order = np.lexsort((tid, frame))
            frame, tid, c, xx, yy = frame[order], tid[order], c[order], xx[order], yy[order]
            bounds = np.searchsorted(frame, np.arange(len(cnt) + 1))
            tabs = []
            for f in range(len(cnt)):
                lo, hi = bounds[f], bounds[f + 1]
                if lo == hi:
                    tabs.append(pd.DataFrame([]))          # "if frame empty, no detections" (:819-821)
                    continue
                tabs.append(pd.DataFrame({'conf': pd.array(c[lo:hi], dtype='Float32'),
                                          'anchor_x': pd.array(xx[lo:hi].astype(np.int64), dtype='Int64'),
                                          'anchor_y': pd.array(yy[lo:hi].astype(np.int64), dtype='Int64')},
                                         index=[f'Axon_{i:0>3}' for i in tid[lo:hi]]))
            self._ided_tables = tabs
        return self._ided_tables

    def _set_ided_from_tables(self, tables):
        """Adopt reference-format per-frame IDed tables (the '_IDed_detections' cache)."""
        cnt, conf, x, y = self._host_dets()
        offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        track = np.full(int(offs[-1]), -1, np.int32)
        for f, t in enumerate(tables):
            if len(t) == 0:
                continue
            tx, ty = t.anchor_x.to_numpy(dtype=np.int64), t.anchor_y.to_numpy(dtype=np.int64)
            for name, ax, ay in zip(t.index, tx, ty):
                k = np.nonzero((x[f, :cnt[f]] == ax) & (y[f, :cnt[f]] == ay))[0][0]   # exact anchor match (:804-808)
                track[offs[f] + k] = _axon_number(name)
        self._track_flat_cache, self._d_track = track, None
        self.n_ids = None                                           # unknown: ids are whatever the cache holds
        self._solved, self._ided_tables = True, list(tables)

    def astar_dists(self):
        """_get_astar_path_distances(_compute_detections_astar_paths()) (AxonDetections.py:526-585,717-752):
        dict '{name}_t:{t:03}-t:{t_bef:03}' -> int array [N_t_bef, N_t], computed on the GPU per pair."""
        cnt = self._host_dets()[0]
        out = {}
        for t in range(len(self)):
            mask = self._mask_dev(t)                       # _get_maskweights(t), AxonDetections.py:557
            for t_bef in range(t - 1, t - (self.P['MCF_MAX_NUM_MISSES'] + 2), -1):
                if t_bef < 0:
                    continue
                na, nb = int(cnt[t_bef]), int(cnt[t])
                lbl = f'{self.dataset.name}_t:{t:0>3}-t:{t_bef:0>3}'
                if na == 0:
                    out[lbl] = np.array([])
                    continue
                D = hp.path_cost(self.d_x[t_bef, :na], self.d_y[t_bef, :na], self.d_x[t, :nb], self.d_y[t, :nb],
                                 self.dataset.sizey, self.dataset.sizex, mask, self.max_px_assoc_dist, self.conn8)
                out[lbl] = D.cpu().numpy()
        return out

    def _length_table_from_dists(self, dists):
        """astar_dists() (int matrices per frame pair, max_px_assoc_dist = none) as the i16 table [F, cap, gaps, cap]
        the arc builder reads (0 = none)."""
        F, cap, gaps = len(self), self.d_x.shape[1], self.P['MCF_MAX_NUM_MISSES'] + 1
        table = np.zeros((F, cap, gaps, cap), np.int16)
        for t in range(F):
            for t_bef in range(t - 1, t - (gaps + 1), -1):
                if t_bef < 0:
                    continue
                D = dists[f'{self.dataset.name}_t:{t:0>3}-t:{t_bef:0>3}']
                if D.ndim == 2 and D.size:
                    table[t_bef, :D.shape[0], t - t_bef - 1, :D.shape[1]] = np.where(D >= self.max_px_assoc_dist, 0, D)
        return torch.from_numpy(table).to(self.device)

    def _length_table_from_paths(self, paths):
        """_get_astar_path_distances (AxonDetections.py:717-752) of a cached path dictionary, as the i16 table
        [F, cap, gaps, cap] axt_build_arcs_from_lengths reads: getnnz() of a path, 0 (= none) for None."""
        cnt = self._host_dets()[0]
        F, cap, gaps = len(self), self.d_x.shape[1], self.P['MCF_MAX_NUM_MISSES'] + 1
        table = np.zeros((F, cap, gaps, cap), np.int16)
        for t in range(F):
            for t_bef in range(t - 1, t - (gaps + 1), -1):
                if t_bef < 0:
                    continue
                rows = paths[f'{self.dataset.name}_t:{t:0>3}-t:{t_bef:0>3}']
                if len(rows) != cnt[t_bef] or any(len(r) != cnt[t] for r in rows):
                    raise ValueError(f'cached paths of t:{t}-t:{t_bef} do not match the detections')
                for i, row in enumerate(rows):
                    table[t_bef, i, t - t_bef - 1, :len(row)] = [0 if p is None else min(p.getnnz(), 32767) for p in row]
        return torch.from_numpy(table).to(self.device)

    def astar_dets_paths(self):
        """The reference's path dictionary (_compute_detections_astar_paths, AxonDetections.py:526-585): per frame pair
        a list over the detections of t_bef of lists over the detections of t of scipy coo matrices [H, W] (bool)
        marking the cells of a shortest path, or None beyond max_px_assoc_dist. On an all-ones mask every monotone
        staircase between the two anchors is a shortest path; this one walks the columns first, then the rows
        (8-connected: the diagonal first, then straight) (which
        of the equally short paths pyastar2d returns is unpinned; lengths are what the tracker
        uses). On a masked grid the GPU search walks back from every target along its distance field
        (axt_path_cells)."""
        from scipy import sparse
        if self.dataset.masked:
            return self._masked_dets_paths()
        return self._open_dets_paths(self.astar_dists())

    def _open_dets_paths(self, dists):
        """The paths of frame pairs on an all-ones mask (closed form): dists = {label: length matrix} of those pairs."""
        from scipy import sparse
        cnt, _, x, y = self._host_dets()
        H, W = self.dataset.sizey, self.dataset.sizex
        out = {}
        for lbl, D in dists.items():
            t, t_bef = (int(v) for v in re.search(r't:(\d+)-t:(\d+)$', lbl).groups())
            rows = []
            for i in range(int(cnt[t_bef])):
                row = []
                for j in range(int(cnt[t])):
                    if D[i, j] >= self.max_px_assoc_dist:
                        row.append(None)
                        continue
                    xa, ya, xb, yb = int(x[t_bef, i]), int(y[t_bef, i]), int(x[t, j]), int(y[t, j])
                    sx, sy = (1 if xb >= xa else -1), (1 if yb >= ya else -1)
                    if self.conn8:                      # diagonal first, then straight: max(|dx|,|dy|) + 1 cells
                        k = min(abs(xb - xa), abs(yb - ya))
                        dx, dy = abs(xb - xa) - k, abs(yb - ya) - k
                        c = np.concatenate([xa + sx * np.arange(k + 1), xa + sx * (k + np.arange(1, dx + 1)),
                                            np.full(dy, xb)]).astype(np.int64)
                        r = np.concatenate([ya + sy * np.arange(k + 1), np.full(dx, ya + sy * k),
                                            ya + sy * (k + np.arange(1, dy + 1))]).astype(np.int64)
                    else:
                        xs = np.arange(xa, xb + sx, sx)
                        ys = np.arange(ya, yb + sy, sy)
                        r = np.concatenate([np.full(len(xs), ya), ys[1:]])
                        c = np.concatenate([xs, np.full(len(ys) - 1, xb)])
                    row.append(sparse.coo_matrix((np.ones(len(r)), (r, c)), (H, W), bool))
                rows.append(row)
            out[lbl] = rows
        return out

    def _masked_dets_paths(self):
        """astar_dets_paths on a masked grid: one exact single-source search per detection of t_bef, then the walk
        back from every detection of t (hotpath.path_cells)."""
        from scipy import sparse
        cnt = self._host_dets()[0]
        H, W = self.dataset.sizey, self.dataset.sizex
        out = {}
        for t in range(len(self)):
            grid = self._mask_dev(t)
            for t_bef in range(t - 1, t - (self.P['MCF_MAX_NUM_MISSES'] + 2), -1):
                if t_bef < 0:
                    continue
                na, nb = int(cnt[t_bef]), int(cnt[t])
                lbl = f'{self.dataset.name}_t:{t:0>3}-t:{t_bef:0>3}'
                if na == 0 or nb == 0:
                    out[lbl] = [[] for _ in range(na)]
                    continue
                if grid is None:
                    # a frame of a time-varying mask that is all ones: the closed-form staircase of the open grid
                    D = hp.path_cost(self.d_x[t_bef, :na], self.d_y[t_bef, :na], self.d_x[t, :nb], self.d_y[t, :nb],
                                     H, W, None, self.max_px_assoc_dist, self.conn8).cpu().numpy()
                    out.update(self._open_dets_paths({lbl: D}))
                    continue
                D, cells = hp.path_cells(self.d_x[t_bef, :na], self.d_y[t_bef, :na], self.d_x[t, :nb], self.d_y[t, :nb],
                                         H, W, grid, self.max_px_assoc_dist, self.conn8)
                D, cells = D.cpu().numpy(), cells.cpu().numpy()
                rows = []
                for i in range(na):
                    row = []
                    for j in range(nb):
                        if D[i, j] >= self.max_px_assoc_dist:
                            row.append(None)
                            continue
                        c = cells[i, j, :D[i, j]]
                        row.append(sparse.coo_matrix((np.ones(len(c)), (c // W, c % W)), (H, W), bool))
                    rows.append(row)
                out[lbl] = rows
        return out

    def search_MCF_params(self, edge_cost_thr_values=(.4, .6, .7, .8, .9, 1, 1.2, 3),
                          entry_exit_cost_values=(.2, .8, .9, 1, 1.1, 2), miss_rate_values=(0.9, 0.6),
                          vis_sim_weight_values=(0, 0.1), conf_capping_method_values=('ceil', 'scale_to_max')):
        """AxonDetections.py:845-922: re-solve the association for every combination of the five tracker parameters
        (same nesting order) and score each against the labelled identities; one row per combination -- the five
        parameters, then mot_metrics.MOTCHALLENGE_METRICS -- written to '{dir}/MCF_params_results.csv' and returned.
        Every solve is the GPU arc build + flow solve of assign_ids (the detections and their appearance
        histograms stay on the device between combinations); the parameters are restored afterwards. On a masked
        grid the path lengths are computed once, exactly and up to max_px_assoc_dist (astar_dists), and reused by every
        combination -- as the reference reuses its path cache."""
        from . import mot_metrics
        if not self.labelled:
            raise ValueError("no labels: call set_groundtruth() first")
        target = self.get_frame_dets('groundtruth', None, libmot=True)
        names = ('edge_cost_thr', 'entry_exit_cost', 'miss_rate', 'vis_sim_weight', 'conf_capping_method')
        keys = ('MCF_EDGE_COST_THR', 'MCF_ENTRY_EXIT_COST', 'MCF_MISS_RATE', 'MCF_VIS_SIM_WEIGHT', 'MCF_CONF_CAPPING_METHOD')
        before = {k: self.P[k] for k in keys}
        results = []
        # masked grid: the exact path lengths once, for every threshold of the grid (the reference reads its path cache
        # in every iteration, :882); all-ones masks have closed-form lengths
        lengths = self._length_table_from_dists(self.astar_dists()) if self.dataset.masked else None
        try:
            for ec in edge_cost_thr_values:
                for eec in entry_exit_cost_values:
                    for mr in miss_rate_values:
                        for vsw in vis_sim_weight_values:
                            for ccm in conf_capping_method_values:
                                self.P.update(dict(zip(keys, (ec, eec, mr, vsw, ccm))))
                                self.assign_ids(_len_table=lengths)
                                pred = self.get_frame_dets('IDed', None, libmot=True) if self._solved else None
                                ev = mot_metrics.compare_to_groundtruth(target, pred, float(self.nms_min_dist) ** 2)
                                results.append(pd.concat([pd.Series((ec, eec, mr, vsw, ccm), names, dtype=object),
                                                          mot_metrics.summarize(ev)]))
        finally:
            self.P.update(before)
        results = pd.concat(results, axis=1).T
        os.makedirs(self.dir, exist_ok=True)
        results.to_csv(f'{self.dir}/MCF_params_results.csv')
        return results

    def _appearance(self):
        """feature_model's histograms of every detection (device tensors hist f32 [F,cap,180], sums f64 [F,cap]),
        computed once from the centre frames (AxonDetections.py:682-685)."""
        if getattr(self, '_hist', None) is None:
            if hasattr(self.dataset, 'make_resident'):
                self.dataset.make_resident()            # (a host-resident timelapse that has not been streamed yet)
            frames, off = self.dataset.frames, 2
            if not isinstance(self.timepoint_subset, range) and getattr(self, '_shard', None) is None:
                # the centre frames of the subset. (The reference hands the tracker get_frame_and_truedets(i) with i the POSITION in
                # the subset, AxonDetections.py:679-685 -- the image of dataset frame i, not of timepoint_subset[i]; only visible
                # with MCF_VIS_SIM_WEIGHT > 0 under a subset, and not reproduced: the crops here belong to the detections.)
                frames, off = frames[[t + 2 for t in self.timepoint_subset]].contiguous(), 0
            self._hist = hp.box_histograms(frames, self.d_x, self.d_y, self.d_count, t_offset=off,
                                           box=self.axon_box_size)
        return self._hist

    def _mask_dev(self, t=None):
        """The mask as a device grid handle (bit rows + connected components), built once per distinct mask. A static
        mask has one grid; with a time-varying mask `t` selects the grid of the paths that end in detection frame t
        (Timelapse.mask_groups: the reference's mask[t], frame-index quirk included unless
        parameters['REPRODUCE_MASK_FRAME_QUIRK'] is False)."""
        ds = self.dataset
        if ds.mask3d is not None:
            grids, index = self._mask_grids()
            if t is None:
                raise ValueError('the mask changes over time: a frame index is needed')
            return grids[index[t]]
        m = ds.mask2d
        if m is None:
            return None
        if getattr(ds, '_grid', None) is None or ds._grid.conn8 != self.conn8:
            ds._grid = hp.Grid(m, self.conn8, self.device)
        return ds._grid

    def _mask_grids(self):
        """Time-varying mask: (grids, index i32 [frames]) -- one device grid per distinct mask, kept with the timelapse."""
        ds = self.dataset
        quirk = bool(self.P.get('REPRODUCE_MASK_FRAME_QUIRK', True))
        key = (self.conn8, quirk)
        if getattr(ds, '_grids', None) is None or ds._grids[0] != key:
            masks, index = ds.mask_groups(quirk)
            ds._grids = (key, [None if m.all() else hp.Grid(m, self.conn8, self.device) for m in masks], index)
        return ds._grids[1], ds._grids[2]

    def _build_arcs_time_varying(self, dmax, units, vis, src_count):
        """Arcs under a mask that changes over time: the paths of a pair are searched on the mask of its LATER frame
        (AxonDetections.py:557), so one pass of the arc builder per distinct mask -- restricted to the source frames that
        have a target frame under that mask -- and of every pass the arcs whose head frame belongs to it; the union,
        ordered by (tail, gap, head) like a single pass. Global numbering and integer costs do not depend on the pass."""
        grids, index = self._mask_grids()
        F, cap = self.d_x.shape
        gaps = len(dmax)
        dev = self.device
        count = self.d_count
        frame_of = torch.repeat_interleave(torch.arange(F, device=dev), count.long())
        index_d = torch.from_numpy(index).to(dev)
        n_det = int(frame_of.numel())
        parts = []
        for g, grid in enumerate(grids):
            heads = np.nonzero(index == g)[0]
            src = np.zeros(F, bool)
            for gap in range(1, gaps + 1):
                src[heads[heads >= gap] - gap] = True
            sc = torch.where(torch.from_numpy(src).to(dev), count, torch.zeros_like(count))
            if src_count is not None:
                sc = torch.minimum(sc, src_count)
            row_ptr, col, length, gap_a, cost = hp.build_arcs(self.d_x, self.d_y, count, self.dataset.sizey, self.dataset.sizex,
                                                              dmax, units, grid, self.max_px_assoc_dist, self.conn8, vis, None, sc)
            tail = torch.repeat_interleave(torch.arange(n_det, device=dev), (row_ptr[1:n_det + 1] - row_ptr[:n_det]))
            keep = index_d[frame_of[col.long()]] == g
            parts.append((tail[keep], col[keep], length[keep], gap_a[keep], cost[keep]))
        tail, col, length, gap_a, cost = (torch.cat([p[i] for p in parts]) for i in range(5))
        # (tail, gap, head): gap and head get fields as wide as their ranges (a gap > 3 no longer spills into the tail)
        order = torch.argsort((tail * (gaps + 1) + gap_a.long()) * max(n_det, 1) + col.long())
        tail, col, length, gap_a, cost = tail[order], col[order], length[order], gap_a[order], cost[order]
        row_ptr = torch.zeros(F * cap + 1, dtype=torch.int64, device=dev)
        row_ptr[1:n_det + 1] = torch.cumsum(torch.bincount(tail, minlength=n_det), 0)
        row_ptr[n_det + 1:] = row_ptr[n_det]
        return row_ptr, col, length, gap_a, cost

    def _assign_IDs_to_detections(self):
        """AxonDetections.py:631-715 with the tracker replaced by axt_build_arcs + axt_mcf_solve
        (parameters['ASSOCIATION'] = 'mcf', the default and the reference's behaviour) or by the
        frame-to-frame Hungarian variant of BASELINE config 3 ('hungarian')."""
        P = self.P
        vis_w = P['MCF_VIS_SIM_WEIGHT']
        dmax, units = _cost_units_on_device(P, self.max_px_assoc_dist, self.device)
        mode = P.get('ASSOCIATION', 'mcf')
        masked = self.dataset.masked
        varying = self.dataset.mask3d is not None
        shard = getattr(self, '_shard', None)           # set by gather_detections(): solve only this rank's frame pairs
        if mode == 'hungarian':
            ctab = None
            if vis_w or varying:
                # the appearance term: the link costs are those of the flow tracker's arcs (axt_build_arcs_vis: the same
                # admission, the same integers), scattered into a dense table per frame pair
                # (likewise a mask that changes over time: one pass of the arc builder per distinct mask)
                vis = None
                if vis_w:
                    hist, hsum = self._appearance()
                    vis = dict(hist=hist, hsum=hsum, weight=vis_w, miss_rate=P['MCF_MISS_RATE'], thr=P['MCF_EDGE_COST_THR'])
                len_table = None
                if masked and max(dmax) - 1 > 250:       # beyond the hot-path search window: the exact lengths (as for 'mcf')
                    len_table = self._length_table_from_dists(self.astar_dists())
                if varying and len_table is None:
                    row_ptr, col, _, gap, cost = self._build_arcs_time_varying(dmax, units, vis, None)
                else:
                    row_ptr, col, _, gap, cost = hp.build_arcs(self.d_x, self.d_y, self.d_count, self.dataset.sizey,
                                                               self.dataset.sizex, dmax, units,
                                                               None if varying else self._mask_dev(),
                                                               self.max_px_assoc_dist, self.conn8, vis, len_table, None)
                ctab = self._hungarian_cost_table(row_ptr, col, gap, cost, len(dmax))
            track, n_tracks = hp.hungarian_assoc(self.d_x, self.d_y, self.d_count, self.dataset.sizey,
                                                 self.dataset.sizex, dmax, units,
                                                 int(np.rint(P['MCF_EDGE_COST_THR'] * 1e6)),
                                                 self.max_px_assoc_dist, self.conn8,
                                                 *(((shard[0], shard[1]), shard[2]) if shard else (None, None)),
                                                 mask=self._mask_dev() if (masked and ctab is None) else None, ctab=ctab)
            self._d_track, self._track_flat_cache = track, None       # host copies are made on first use only
            self._n_ids, self._n_tracks_dev, self.mcf_total_cost = None, n_tracks, None      # (the count stays on the device)
            return True
        if mode != 'mcf':
            raise ValueError(f"parameters['ASSOCIATION'] must be 'mcf' or 'hungarian', got {mode!r}")
        obs = hp.obs_costs(self.d_conf, self.d_count, P['MCF_CONF_CAPPING_METHOD'], P['MCF_MAX_CONF_COST'])
        vis = None
        if vis_w:
            hist, hsum = self._appearance()
            vis = dict(hist=hist, hsum=hsum, weight=vis_w, miss_rate=P['MCF_MISS_RATE'], thr=P['MCF_EDGE_COST_THR'])
        len_table = getattr(self, '_len_table', None)
        if len_table is None and masked and max(dmax) - 1 > 250:
            # the hot-path searches of the arc builder cover paths of up to 251 cells (the deployed threshold admits
            # exactly that); wider thresholds take the exact, slower search over the whole range
            len_table = self._length_table_from_dists(self.astar_dists())
        src_count = None
        if shard is not None and (masked or vis is not None or len_table is not None):
            # frame-sharded: this rank builds the arc rows of its own frames, one all-gather joins them (the path
            # searches of a masked grid / the appearance distances are the expensive part of the association and shard
            # with the frames). On an all-ones mask the arcs are closed-form and cheaper to rebuild on every rank than to
            # exchange: the all-gather of the detections is then the ONLY collective before the replicated flow solve.
            src_count = torch.zeros_like(self.d_count)
            src_count[shard[0]:shard[1]] = self.d_count[shard[0]:shard[1]]
        if varying and len_table is None:
            row_ptr, col, length, gap, cost = self._build_arcs_time_varying(dmax, units, vis, src_count)
        else:
            row_ptr, col, length, gap, cost = hp.build_arcs(self.d_x, self.d_y, self.d_count, self.dataset.sizey,
                                                            self.dataset.sizex, dmax, units,
                                                            None if varying else self._mask_dev(),
                                                            self.max_px_assoc_dist, self.conn8, vis, len_table, src_count)
        cnt, conf, x, y = self._host_dets()
        if src_count is not None:
            from . import sharded
            row_ptr, col, length, gap, cost = sharded.all_gather_arcs(row_ptr, col, length, gap, cost, int(cnt.sum()),
                                                                      shard[2])
        offs = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        n_det = int(offs[-1])
        # the integer node costs (axt_arc_cost_int: round(cost * 1e6) << 16 | identity hash) on the device, beside the arcs' (the
        # numpy version of the same arithmetic took 57 ms for config 4's 319 k detections); entry / exit depend on the count
        # and the price only and are kept
        valid = torch.arange(conf.shape[1], device=self.device)[None, :] < self.d_count[:, None]
        k = torch.arange(n_det, dtype=torch.int64, device=self.device)
        obs_int = _arc_cost_int_torch(torch.round(obs[valid] * 1e6).to(torch.int64), 2, k, 0)
        ee = float(P['MCF_ENTRY_EXIT_COST'])
        key = (n_det, ee, str(self.device))
        if _ENTRY_EXIT_CACHE.get('key') != key:
            units = torch.full((n_det,), int(np.rint(ee * 1e6)), dtype=torch.int64, device=self.device)
            _ENTRY_EXIT_CACHE.update(key=key, entry=_arc_cost_int_torch(units, 0, k, 0).cpu().numpy(),
                                     exit=_arc_cost_int_torch(units, 1, k, 0).cpu().numpy())
        entry_int, exit_int = _ENTRY_EXIT_CACHE['entry'], _ENTRY_EXIT_CACHE['exit']
        obs_int, row_ptr_h, col_h, cost_h = hp.to_host(obs_int, row_ptr[:n_det + 1], col, cost)       # (pinned staging: 120 MB at config 4)
        net = (obs_int, entry_int, exit_int, row_ptr_h, col_h, cost_h)
        # parameters['MCF_CERTIFICATE'] (not a key of the reference): also return the node potentials that prove the optimum
        # (axt_mcf_solve_duals) and keep them, with the network they refer to, in self.mcf_certificate
        want_cert = bool(P.get('MCF_CERTIFICATE', False))
        self.mcf_certificate = None
        if shard is not None and P.get('MCF_SHARDED_SOLVE', True):
            # frame-sharded: every rank solves its run of time blocks, one all-gather joins the runs (sharded.solve_flow)
            from . import sharded
            res = sharded.solve_flow(*net, P['MCF_MIN_FLOW'], P['MCF_MAX_FLOW'], shard[2], self.device, duals=want_cert)
        else:
            res = hp.mcf_solve(*net, P['MCF_MIN_FLOW'], P['MCF_MAX_FLOW'], duals=want_cert)
        if res is not None and want_cert:
            self.mcf_certificate = dict(obs=net[0].copy(), entry=net[1], exit=net[2], row_ptr=net[3].copy(), col=net[4].copy(),
                                        cost=net[5].copy(), next=res[0], track=res[1], total_cost=res[3], potentials=res[4],
                                        min_flow=P['MCF_MIN_FLOW'], max_flow=P['MCF_MAX_FLOW'])
            res = res[:4]
        if res is None:
            print('Could not solve the graph for identity association; -> no IDed detections. Try narrowing '
                  'expected identities by updating parameters[`MCF_MIN_FLOW`, `MCF_MAX_FLOW`]. '
                  f"Currently: {P['MCF_MIN_FLOW']} to {P['MCF_MAX_FLOW']}.")
            return False
        nxt, track, n_tracks, total = res
        self.mcf_total_cost, self.n_ids = total, n_tracks
        self._track_flat_cache, self._d_track = track, None
        return True

    def _hungarian_cost_table(self, row_ptr, col, gap, cost, gaps):
        """Arcs (CSR by global tail index, global head indices, integer costs) -> i64 [F, cap, gaps, cap] with
        hp.HUNGARIAN_NO_LINK where there is no arc: what axt_hungarian_pairs_costs reads. All on the device."""
        F, cap = self.d_x.shape
        if F * cap * gaps * cap * 8 > 4 << 30:
            raise NotImplementedError(f'the cost table of {F} frames x {cap} detection slots would take '
                                      f'{F * cap * gaps * cap * 8 / 2 ** 30:.1f} GiB')
        dev = self.device
        offs = torch.zeros((F + 1,), dtype=torch.int64, device=dev)
        offs[1:] = torch.cumsum(self.d_count.long(), 0)
        n_det = int(offs[-1].item())
        deg = (row_ptr[1:n_det + 1] - row_ptr[:n_det]).long()
        tail = torch.repeat_interleave(torch.arange(n_det, device=dev), deg, output_size=len(col))
        ft = torch.searchsorted(offs, tail, right=True) - 1
        head = col.long()
        fb = torch.searchsorted(offs, head, right=True) - 1
        tab = torch.full((F, cap, gaps, cap), hp.HUNGARIAN_NO_LINK, dtype=torch.int64, device=dev)
        tab[ft, tail - offs[ft], gap.long() - 1, head - offs[fb]] = cost
        return tab

    @property
    def _offs(self):
        """Start of every frame in the flat (frame-major) detection numbering, i64 [F+1]."""
        return np.concatenate([[0], np.cumsum(self._host_dets()[0])]).astype(np.int64)

    @property
    def _track_flat(self):
        """Trajectory id of every detection in flat numbering on the host, i32 [n_det] (-1: none)."""
        if getattr(self, '_track_flat_cache', None) is None:
            cnt = self._host_dets()[0]
            track_h = self._d_track.cpu().numpy()
            self._track_flat_cache = track_h[np.arange(track_h.shape[1])[None, :] < cnt[:, None]]
        return self._track_flat_cache

    def ided_arrays(self):
        """(frame i32, id i32, conf f32, x i32, y i32) of every IDed detection, frame-major."""
        cnt, conf, x, y = self._host_dets()
        frame_of = np.repeat(np.arange(len(cnt)), cnt)
        k = np.arange(len(frame_of))
        idx_in = k - self._offs[frame_of]
        track = self._track_flat
        sel = track >= 0
        return (frame_of[sel], track[sel], conf[frame_of[sel], idx_in[sel]], x[frame_of[sel], idx_in[sel]],
                y[frame_of[sel], idx_in[sel]])

    def _agg_all_IDed_dets(self):
        """AxonDetections.py:825-842, including (by default) its frame-label quirk: labels are
        column_position//3, so frames after one without IDed detections are labelled one too low.

        The table is dense [n_ids, 3*frames] f64 with NaN where an axon is absent -- its size grows with
        frames x ids, so it is filled on the GPU (axt_ided_table) and copied once into pinned host memory,
        which the DataFrame then wraps without another copy."""
        track = self._track_dev()                                   # i32 [F,cap], -1 = no ID / empty slot
        shard = getattr(self, '_shard', None)
        if shard is not None:
            return self._ided_block(track, shard[0], shard[1])
        key = (int(track.shape[0]), int(track.shape[1]))
        if self._n_ids is None and self._n_tracks_dev is not None and key in _IDS_GUESS:
            # The number of identities is still on the device, and fetching it first would hold the host -- and with it the
            # launches below -- until the GPU has drained the pass. The table is built for the count of the previous pass over
            # a timelapse of this shape (+ a margin) instead, the count travels with it, and rows beyond it are not shown; only
            # if the guess turns out too small is the table built again.
            rows = _IDS_GUESS[key]
            got = _pinned_count()
            got.copy_(self._n_tracks_dev, non_blocking=True)
            vals, wait = hp.ided_table(track, self.d_conf, self.d_x, self.d_y, self.d_count, rows, self.reproduce_label_quirk, None, rows)
            wait()
            n = self._n_ids = int(got[0])
            _IDS_GUESS[key] = (n // 64 + 2) * 64
            if n <= rows:
                return pd.DataFrame(vals[:n], index=_axon_index(np.arange(n)), columns=_ided_columns(len(self)), copy=False)
        n_ids, ids, id_row = self.n_ids, None, None
        if n_ids is not None:
            _IDS_GUESS[key] = (n_ids // 64 + 2) * 64
        if n_ids is None:                                           # adopted from a cache: ids may have gaps
            uniq = torch.unique(track[track >= 0])
            ids = uniq.cpu().numpy()
            n_ids = int(ids[-1]) + 1 if len(ids) else 0
            id_row = torch.full((max(n_ids, 1),), -1, dtype=torch.int32, device=self.device)
            id_row[uniq.long()] = torch.arange(len(ids), dtype=torch.int32, device=self.device)
        vals, wait = hp.ided_table(track, self.d_conf, self.d_x, self.d_y, self.d_count, n_ids, self.reproduce_label_quirk,
                                   id_row, None if ids is None else len(ids))
        df = pd.DataFrame(vals, index=_axon_index(np.arange(n_ids) if ids is None else ids),
                          columns=_ided_columns(len(self)), copy=False)          # built while the copy is in flight
        wait()
        return df

    def _ided_block(self, track, a, b):
        """Frame-sharded runs: the dense table of a timelapse grows with frames x identities, so every rank
        materialises only its own frames [a, b) of it -- the identities alive there (rows) x those frames (columns,
        labelled with their true frame index, no label quirk yet). sharded.assemble_ided_dets_all() joins the blocks
        into exactly the table a single process builds (quirk included); the work and the PCIe traffic per rank stay
        constant as ranks are added."""
        tr = track[a:b]
        alive = torch.unique(tr[tr >= 0])
        id_row = torch.full((max(int(self.n_ids or 0), 1),), -1, dtype=torch.int32, device=self.device)
        id_row[alive.long()] = torch.arange(len(alive), dtype=torch.int32, device=self.device)
        vals, wait = hp.ided_table(tr, self.d_conf[a:b], self.d_x[a:b], self.d_y[a:b], self.d_count[a:b], int(self.n_ids or 0),
                                   False, id_row, len(alive))
        cols = pd.MultiIndex.from_product([range(a, b), ['anchor_x', 'anchor_y', 'conf']], names=('frameID', 'detInfo'))
        self.IDed_dets_block = (a, b)
        df = pd.DataFrame(vals, index=_axon_index(alive.cpu().numpy()), columns=cols, copy=False)
        wait()
        return df

    def _track_dev(self):
        """Trajectory id of every detection slot on the device, i32 [F,cap] (-1: none)."""
        t = getattr(self, '_d_track', None)
        if t is None:
            cnt = self._host_dets()[0]
            cap = self.d_conf.shape[1]
            full = np.full((len(cnt), cap), -1, np.int32)
            valid = np.arange(cap)[None, :] < cnt[:, None]
            full[valid] = self._track_flat_cache
            t = self._d_track = torch.from_numpy(full).to(self.device)
        return t


_COLUMNS_CACHE = {}
_UNITS_CACHE = {}
_ENTRY_EXIT_CACHE = {}          # the integer entry / exit costs of the last (detection count, price, device)
_IDS_GUESS = {}                 # (frames, capacity) -> rows to build IDed_dets_all for before the count is known
_COUNT_BUF = []


def _pinned_count():
    if not _COUNT_BUF:
        _COUNT_BUF.append(torch.zeros((1,), dtype=torch.int32, pin_memory=True))
    return _COUNT_BUF[0]
_THRS_CACHE = {}


def _conf_thresholds(conf_thr):
    """all_conf_thrs (AxonDetections.py:65-66): the thirteen thresholds of the detection metrics, read-only, per BBOX_THRESHOLD."""
    if conf_thr not in _THRS_CACHE:
        a = np.sort(np.append(np.arange(0.55, 1, .04), conf_thr)).round(2)
        a.setflags(write=False)
        _THRS_CACHE[conf_thr] = a
    return _THRS_CACHE[conf_thr]


def _cost_units_on_device(P, max_px, device):
    """(dmax i32 [gaps], integer cost table i64 [gaps, max_px+1] on `device`) of the transition model for these
    parameters. Cached: the table is a pure function of four parameters, and uploading it from pageable memory in
    the middle of a pass would make the host wait for the detector."""
    key = (P['MCF_VIS_SIM_WEIGHT'], P['MCF_MAX_NUM_MISSES'], P['MCF_MISS_RATE'], P['MCF_EDGE_COST_THR'], max_px, str(device))
    if key not in _UNITS_CACHE:
        table, dmax = transition_cost_table(P, max_px, vis_sim=1.0 if P['MCF_VIS_SIM_WEIGHT'] else 0.0)
        units = np.where(np.isfinite(table), np.rint(table * 1e6), 0).astype(np.int64)
        _UNITS_CACHE[key] = (dmax, torch.from_numpy(units).to(device))
    return _UNITS_CACHE[key]


def _ided_columns(F):
    """MultiIndex (frameID, detInfo) of IDed_dets_all; building it costs more than the rest of the table."""
    if F not in _COLUMNS_CACHE:
        _COLUMNS_CACHE[F] = pd.MultiIndex.from_product([range(F), ['anchor_x', 'anchor_y', 'conf']],
                                                       names=('frameID', 'detInfo'))
    return _COLUMNS_CACHE[F]


_AXON_NAMES = [f'Axon_{i:0>3}' for i in range(2048)]


def _axon_number(name):
    """'Axon_012' -> 12, 'Axon_1234' -> 1234 (names are zero-padded to three digits, not truncated)."""
    return int(str(name).split('_')[-1])


def _axon_index(ids):
    names = [_AXON_NAMES[i] if i < 2048 else f'Axon_{i:0>3}' for i in ids]
    return pd.Index(names, name='axonID')


def _splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _lsr(x, s):
    """logical right shift of an int64 tensor holding uint64 bits"""
    return (x >> s) & ((1 << (64 - s)) - 1)


def _arc_cost_int_torch(units, kind, a, b):
    """axt_arc_cost_int on int64 device tensors (units = round(cost*1e6) already)."""
    def s64(v):                                   # python int -> signed 64-bit two's complement
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v
    x = (a << 30) ^ b ^ s64(kind << 60)
    x = x + s64(0x9E3779B97F4A7C15)
    x = (x ^ _lsr(x, 30)) * s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * s64(0x94D049BB133111EB)
    x = x ^ _lsr(x, 31)
    return units * 65536 + (x & 0xFFFF)


def _arc_cost_int_vec(cost, kind, a, b):
    """Vectorised axt_arc_cost_int (include/axtrack_hip.h)."""
    with np.errstate(over='ignore'):
        key = (np.uint64(kind) << np.uint64(60)) ^ (np.asarray(a, np.uint64) << np.uint64(30)) ^ np.asarray(b, np.uint64)
        pert = (_splitmix64(key) & np.uint64(0xFFFF)).astype(np.int64)
    return np.rint(np.asarray(cost, np.float64) * 1e6).astype(np.int64) * 65536 + pert
