"""Input container of the hot path: the timelapse as ONE dense f32 tensor in HBM.

Mirrors the attributes of the reference's Timelapse (axtrack/Timelapse.py) that the inference
path reads -- name, sizet/sizey/sizex, tilesize, mask, len() -- but keeps only what the
detector consumes: channel 0 of `X` (Timelapse.py:426-433; the two motion channels are all
zeros for USE_MOTION_DATA='exclude', :366-367), context-padded: T_all = sizet + 2*context.
"""
import os

import numpy as np
import torch


class Timelapse:
    def __init__(self, frames, name='timelapse', mask=None, temporal_context=2, tilesize=512,
                 device='cuda:0', pixelsize=None, dt=None, incubation_time=None):
        """frames: preprocessed f32 [T_all,H,W] (numpy or torch); mask: bool [H,W] or None (all ones)."""
        if temporal_context != 2:
            raise ValueError('the deployed detector has 5 input channels: temporal_context must be 2')
        f = torch.as_tensor(frames)
        if f.dim() != 3:
            raise ValueError(f'frames must be [T_all,H,W], got {tuple(f.shape)}')
        if f.shape[0] < 2 * temporal_context + 1:
            raise ValueError('need at least 5 frames (2 context frames either side of one detection frame)')
        self.frames = f.to(device=device, dtype=torch.float32).contiguous()
        self.name = name
        self.temporal_context = temporal_context
        self.tilesize = tilesize
        self.sizet = self.frames.shape[0] - 2 * temporal_context
        self.sizey, self.sizex = int(self.frames.shape[1]), int(self.frames.shape[2])
        self.ytiles, self.xtiles = -(-self.sizey // tilesize), -(-self.sizex // tilesize)
        # mask: one for the whole timelapse, or one per input frame (Timelapse.py:210-217 stacks a 2-D mask to [T,H,W];
        # AxonDetections._get_maskweights(t) then reads mask[t], AxonDetections.py:587-598). mask2d is the static
        # mask (None = all ones); mask3d is set only when the mask really changes over time.
        self.mask2d, self.mask3d = None, None
        if mask is not None:
            mask = np.asarray(mask).astype(bool)
            if mask.ndim == 3:
                if mask.shape != (self.frames.shape[0], self.sizey, self.sizex):
                    raise ValueError(f'a time-varying mask must be [T_all,H,W] = {tuple(self.frames.shape)}, got {mask.shape}')
                if (mask == mask[0]).all():
                    mask = mask[0]
            if mask.ndim == 2:
                if mask.shape != (self.sizey, self.sizex):
                    raise ValueError('mask must be [H,W] or [T_all,H,W]')
                self.mask2d = None if mask.all() else mask
            elif mask.ndim == 3:
                self.mask3d = mask
            else:
                raise ValueError('mask must be [H,W] or [T_all,H,W]')
        self.pixelsize, self.dt, self.incubation_time = pixelsize, dt, incubation_time
        self.timepoints = np.arange(temporal_context, temporal_context + self.sizet)

    def __len__(self):
        return self.sizet

    @property
    def masked(self):
        return self.mask2d is not None or self.mask3d is not None

    def mask_groups(self, quirk=True):
        """Time-varying masks: (list of distinct [H,W] masks, index i32 [sizet]) -- which mask the reference searches the
        paths on that END in detection frame t. With quirk=True (default) that is mask[t], the mask of INPUT frame t:
        _get_maskweights(t) indexes the context-padded frame list with the detection-frame index (AxonDetections.py:557,
        598; Timelapse.py:408), two frames before the one detection frame t shows. quirk=False takes mask[t + context]."""
        m = self.mask3d
        keys, masks, index = {}, [], np.zeros(self.sizet, np.int32)
        for t in range(self.sizet):
            f = m[t if quirk else t + self.temporal_context]
            k = f.tobytes()
            if k not in keys:
                keys[k] = len(masks)
                masks.append(f)
            index[t] = keys[k]
        return masks, index

    def to_cache(self, directory):
        """'{name}_dataset_cached.pkl' (Timelapse._caching, Timelapse.py:435-449): the reference pickles its whole
        __dict__ (sparse tensors of all three channels); this writes the same file name with what the hot path keeps --
        the preprocessed frames, the mask and the metadata."""
        import os
        import pickle
        os.makedirs(directory, exist_ok=True)
        self.make_resident()
        d = dict(_axtrack_amd_cache=1, name=self.name, frames=self.frames.cpu().numpy(),
                 mask=self.mask3d if self.mask3d is not None else self.mask2d, temporal_context=self.temporal_context,
                 tilesize=self.tilesize, pixelsize=self.pixelsize, dt=self.dt, incubation_time=self.incubation_time)
        with open(f'{directory}/{self.name}_dataset_cached.pkl', 'wb') as file:
            pickle.dump(d, file, protocol=4)

    @classmethod
    def from_cache(cls, directory, name, device='cuda:0'):
        """Read '{name}_dataset_cached.pkl': this package's own format, or a file the reference wrote (its __dict__: `X` a
        sparse tensor [T_all, 3, H, W] whose channel 0 is the preprocessed image, `mask` a list of scipy coo matrices)."""
        import os
        import pickle
        fname = f'{directory}/{name}_dataset_cached.pkl'
        assert os.path.exists(fname), f'\n\nNo cached dataset found: {fname}'
        with open(fname, 'rb') as file:
            d = pickle.load(file)
        if d.get('_axtrack_amd_cache'):
            return cls(d['frames'], name=d['name'], mask=d['mask'], temporal_context=d['temporal_context'],
                       tilesize=d['tilesize'], device=device, pixelsize=d['pixelsize'], dt=d['dt'],
                       incubation_time=d['incubation_time'])
        X = d['X']
        X = X.to_dense() if X.is_sparse else X
        mask = np.stack([np.asarray(m.todense()) if hasattr(m, 'todense') else np.asarray(m) for m in d['mask']]).astype(bool)
        return cls(X[:, 0].contiguous(), name=d.get('name', name), mask=mask, temporal_context=d.get('temporal_context', 2),
                   tilesize=d.get('tilesize', 512), device=device, pixelsize=d.get('pixelsize'), dt=d.get('dt'),
                   incubation_time=d.get('incubation_time'))

    # ------------------------------------------------------------------ host-resident input (Timelapse.py:205-326 + 492-566)
    @classmethod
    def from_host_u16(cls, raw, name='timelapse', mask=None, offset=121, clip=55, log_correct=True, scale=0.015176106,
                      chunk_frames=96, device='cuda:0', **kw):
        """A timelapse whose raw uint16 frames [T_all,H,W] still sit in HOST memory (pinned here if they are not), as the
        reference's inference() starts from (its Timelapse is a host object; construct_tiles copies to the device,
        Timelapse.py:492-566). Nothing is copied yet: AxonDetections.detect_dataset() streams the frames in chunks of
        `chunk_frames` -- H2D copies on a second stream into two staging buffers, the fused preprocessing pass
        (axt_preprocess_u16) into this object's frame buffer and the CNN of the detection frames a chunk completes, so the
        copy of chunk k + 1 runs beside the kernels of chunk k. Afterwards the object is an ordinary resident Timelapse.
        mask: [H,W] or None (a mask per frame needs the resident path: prepare_input_data)."""
        a = raw if isinstance(raw, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(raw)).view(np.int16))
        if a.dtype == torch.uint16:
            a = a.view(torch.int16)
        if a.dtype != torch.int16 or a.dim() != 3:
            raise TypeError('raw timelapses are uint16 [T_all,H,W]')
        if mask is not None and np.asarray(mask).ndim != 2:
            raise ValueError('from_host_u16 takes a static [H,W] mask; a mask per frame goes through prepare_input_data')
        self = cls.__new__(cls)
        T, H, W = a.shape
        dev = torch.device(device)
        frames = torch.empty((T, H, W), dtype=torch.float32, device=dev)
        cls.__init__(self, frames, name=name, mask=mask, device=device, **kw)
        self.frames = frames                                            # (the constructor keeps the same storage)
        self._host_raw = a if a.is_pinned() else a.pin_memory()
        off = 0.0 if not offset else (offset / 2 ** 16 if isinstance(offset, int) else float(offset))
        lo = 0.0 if not clip else (clip / 2 ** 16 if isinstance(clip, int) else float(clip))
        m = None if mask is None else torch.from_numpy(np.ascontiguousarray(np.asarray(mask).astype(np.uint8))).to(dev)
        self._pre = (m, off, lo, bool(log_correct), float(scale))
        self._chunk = max(int(chunk_frames), 2 * self.temporal_context + 1)
        self._pending = True
        return self

    def stream_chunks(self):
        """Generator over the chunks of a host-resident timelapse. All H2D copies are enqueued at once on a copy stream, in
        pieces of 16 frames into a device buffer for the raw timelapse (2 bytes per pixel; no staging buffer to recycle), each
        followed by an event. The compute (current) stream then takes the frames in chunks that GROW -- 16, 32, 48, 64, 80 frames,
        then `chunk_frames` -- so that the first kernels start after ~0.17 ms of copying while later chunks are large enough
        for full launches (PCIe delivers a 512x512 frame in 10.6 us, the detector needs ~14 us for it: the copies stay ahead
        of a schedule that grows no faster than that ratio allows; profiles/r03j_trace). Per chunk: wait for its last piece, the fused preprocessing pass into the frame buffer, the
        tile occupancy; then yields (first frame, one past last frame, occupancy bytes so far) so that the caller can
        enqueue the work the chunk completes. Leaves the frames resident."""
        from . import hotpath as hp
        dev = self.frames.device
        T, H, W = self.frames.shape
        piece = 16
        ramp = tuple(int(v) for v in os.environ.get('AXT_STREAM_RAMP', '16,32,48,64,80').split(','))      # tuning knob (bench experiments)
        copy_stream = _copy_stream(dev)
        compute = torch.cuda.current_stream(dev)
        copy_stream.wait_stream(compute)                 # (the stream is shared between timelapses: start behind what is queued)
        d_raw = torch.empty((T, H, W), dtype=torch.int16, device=dev)
        landed = []

        def enqueue_copies(upto):                       # pieces covering frames [.., upto)
            with torch.cuda.stream(copy_stream):
                while len(landed) * piece < min(upto, T):
                    a0 = len(landed) * piece
                    d_raw[a0:a0 + piece].copy_(self._host_raw[a0:a0 + piece], non_blocking=True)
                    landed.append(copy_stream.record_event())
        m, off, lo, logc, scale = self._pre
        a, k = 0, 0
        sizes = lambda kk: min(ramp[kk], self._chunk) if kk < len(ramp) else self._chunk
        while a < T:
            n = sizes(k)
            b = min(a + n, T)
            if T - b < piece:
                b = T
            # the copies of this chunk and of the next one are in the copy queue before this chunk's kernels are enqueued (all
            # of them up front would keep the host busy for ~0.5 ms before the first kernel launch)
            enqueue_copies(b + sizes(k + 1) + piece)
            compute.wait_event(landed[(b - 1) // piece])
            hp.preprocess_u16(d_raw[a:b], m, off, lo, logc, scale, out=self.frames[a:b])
            occ = None
            if b == T:
                # the kept-tile list (one pass over the finished frames) goes to the host on the copy stream, behind this
                # point of the compute stream only: the host does not wait for the CNN launches the caller enqueues for
                # this last chunk
                occ = hp.tile_occupancy_bytes(self.frames)
                ready = compute.record_event()
                self._occ_host = _pinned_bytes(int(occ.numel()))
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(ready)
                    self._occ_host.copy_(occ, non_blocking=True)
                    self._occ_done = copy_stream.record_event()
            yield a, b, occ
            a, k = b, k + 1
        d_raw.record_stream(copy_stream)
        self._pending = False
        self._host_raw = None

    def make_resident(self):
        """A host-resident timelapse (from_host_u16) holds an UNINITIALISED frame buffer until its chunks have been streamed:
        everything that reads `frames` other than detect_dataset's own streaming loop (the tile list, the dataset cache, the
        appearance histograms, the occupancy all-reduce) calls this first. A no-op for a resident timelapse."""
        if getattr(self, '_pending', False):
            for _ in self.stream_chunks():
                pass
        return self

    @property
    def tile_yx(self):
        """Row-major list of the (tile_row, tile_col) that hold a non-zero pixel at some time point: the reference's
        tile_info, which construct_tiles computes once when the dataset is built (Timelapse.py:551-558), not in
        inference(). Computed on the GPU (axt_tile_occupancy) on first use and kept with the timelapse."""
        if getattr(self, '_tile_yx', None) is None:
            from . import hotpath as hp
            self.make_resident()
            self._tile_yx = hp.tile_occupancy(self.frames)
        return self._tile_yx

    def sync_tile_occupancy(self, group=None):
        """Frame-sharded runs: this object holds one rank's block of frames, but the reference decides which tiles
        are empty over the WHOLE timelapse (non_empty_tiles.any over all time points, Timelapse.py:551-558). One MAX
        all-reduce of the occupancy bytes gives every rank the timelapse-wide tile list -- and with it the same
        n_tiles, hence the same detection capacity, which the all-gather of the detections relies on."""
        import torch.distributed as dist
        from . import hotpath as hp
        self.make_resident()
        occ = hp.tile_occupancy_bytes(self.frames)
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == 'gloo':
                occ_h = occ.cpu()
                dist.all_reduce(occ_h, op=dist.ReduceOp.MAX, group=group)
                occ = occ_h
            else:
                dist.all_reduce(occ, op=dist.ReduceOp.MAX, group=group)
        self._tile_yx = hp.tile_list(occ, self.sizey, self.sizex)
        return self._tile_yx

    @property
    def device(self):
        return self.frames.device


_PINNED = {}
_COPY_STREAMS = {}


def _copy_stream(dev):
    """One copy stream per device, created once (creating a stream costs as much as a small kernel)."""
    key = str(dev)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _COPY_STREAMS[key]


def _pinned_bytes(n):
    """A pinned host byte buffer of n bytes, allocated once per size (pinning costs ~100 us, more than the copy it serves)."""
    if n not in _PINNED:
        _PINNED[n] = torch.empty((n,), dtype=torch.uint8, pin_memory=True)
    return _PINNED[n]


def preprocess(imseq, mask=None, offset=121, clip=55, log_correct=True, scale=0.015176106, device='cuda:0', pad=None):
    """Dense preprocessing of a raw uint16 timelapse as Timelapse._read_tiff / _clip_image_values /
    _log_adjust_image / _standardize do it (Timelapse.py:205-326), as one fused HIP pass
    (axt_preprocess_u16): u16 -> f32 in [0,1], mask, subtract offset/2^16 and clamp at 0, zero below
    clip/2^16, log2(1+x), divide by the train-set std. `img_as_float32` and `adjust_log` are skimage
    functions that are absent here: their arithmetic (x * (1/65535), log2(1+x)) is restated from the
    published skimage 0.18 behaviour, PARITY UNPINNED (a "next" row).
    mask: [H,W] or one per frame [T,H,W] (Timelapse.py:210-217). pad: None or (top, right, bottom, left) -- zero
    margins added after masking and offsetting (Timelapse.py:224-234); zero stays zero through the clip, the log and
    the scaling, so the margins are added to the finished frames. Returns f32 [T, H + top + bottom, W + left + right]."""
    from . import hotpath as hp
    a = np.asarray(imseq)
    if a.dtype != np.uint16:
        raise TypeError(f'raw timelapses are uint16 (got {a.dtype}); pass preprocessed float32 frames to Timelapse directly')
    raw = torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(device)
    off = 0.0 if not offset else (offset / 2 ** 16 if isinstance(offset, int) else float(offset))
    lo = 0.0 if not clip else (clip / 2 ** 16 if isinstance(clip, int) else float(clip))
    m = None
    if mask is not None:
        mk = np.asarray(mask).astype(bool)
        if mk.ndim == 3:                      # a mask per frame: zero the raw counts, which is what masking the floats does
            if mk.shape != a.shape:
                raise ValueError(f'a time-varying mask must match the timelapse {a.shape}, got {mk.shape}')
            raw = raw * torch.from_numpy(mk).to(device=device, dtype=torch.int16)
        else:
            m = torch.from_numpy(np.ascontiguousarray(mk.astype(np.uint8)))
    out = hp.preprocess_u16(raw.contiguous(), m, off, lo, bool(log_correct), float(scale))
    if pad is not None and any(pad):
        top, right, bottom, left = (int(v) for v in pad)
        out = torch.nn.functional.pad(out, (left, right, top, bottom)).contiguous()
    return out


def pad_mask(mask, pad, shape):
    """The mask of a padded timelapse (Timelapse.py:231-234): zeros in the margins; an absent mask becomes the all-ones
    mask of the unpadded frame, so the margins are off the mask for the path searches exactly as in the reference."""
    top, right, bottom, left = (int(v) for v in pad)
    T, H, W = shape
    m = np.ones((H, W), bool) if mask is None else np.asarray(mask).astype(bool)
    width = ((top, bottom), (left, right))
    return np.pad(m, width if m.ndim == 2 else ((0, 0),) + width)
