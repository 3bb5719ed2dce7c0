"""Input container of the hot path: the timelapse as ONE dense f32 tensor in HBM.

Mirrors the attributes of the reference's Timelapse (axtrack/Timelapse.py) that the inference
path reads -- name, sizet/sizey/sizex, tilesize, mask, len() -- but keeps only what the
detector consumes: channel 0 of `X` (Timelapse.py:426-433; the two motion channels are all
zeros for USE_MOTION_DATA='exclude', :366-367), context-padded: T_all = sizet + 2*context.
"""
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
        if mask is not None:
            mask = np.asarray(mask).astype(bool)
            if mask.shape != (self.sizey, self.sizex):
                raise ValueError('mask must be [H,W]')
            if mask.all():
                mask = None
        self.mask2d = mask
        self.pixelsize, self.dt, self.incubation_time = pixelsize, dt, incubation_time
        self.timepoints = np.arange(temporal_context, temporal_context + self.sizet)

    def __len__(self):
        return self.sizet

    @property
    def tile_yx(self):
        """Row-major list of the (tile_row, tile_col) that hold a non-zero pixel at some time point: the reference's
        tile_info, which construct_tiles computes once when the dataset is built (Timelapse.py:551-558), not in
        inference(). Computed on the GPU (axt_tile_occupancy) on first use and kept with the timelapse."""
        if getattr(self, '_tile_yx', None) is None:
            from . import hotpath as hp
            self._tile_yx = hp.tile_occupancy(self.frames)
        return self._tile_yx

    def sync_tile_occupancy(self, group=None):
        """Frame-sharded runs: this object holds one rank's block of frames, but the reference decides which tiles
        are empty over the WHOLE timelapse (non_empty_tiles.any over all time points, Timelapse.py:551-558). One MAX
        all-reduce of the occupancy bytes gives every rank the timelapse-wide tile list -- and with it the same
        n_tiles, hence the same detection capacity, which the all-gather of the detections relies on."""
        import torch.distributed as dist
        from . import hotpath as hp
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


def preprocess(imseq, mask=None, offset=121, clip=55, log_correct=True, scale=0.015176106, device='cuda:0'):
    """Dense preprocessing of a raw uint16 timelapse as Timelapse._read_tiff / _clip_image_values /
    _log_adjust_image / _standardize do it (Timelapse.py:205-326), as one fused HIP pass
    (axt_preprocess_u16): u16 -> f32 in [0,1], mask, subtract offset/2^16 and clamp at 0, zero below
    clip/2^16, log2(1+x), divide by the train-set std. `img_as_float32` and `adjust_log` are skimage
    functions that are absent here: their arithmetic (x * (1/65535), log2(1+x)) is restated from the
    published skimage 0.18 behaviour, PARITY UNPINNED (a "next" row)."""
    from . import hotpath as hp
    a = np.asarray(imseq)
    if a.dtype != np.uint16:
        raise TypeError(f'raw timelapses are uint16 (got {a.dtype}); pass preprocessed float32 frames to Timelapse directly')
    raw = torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(device)
    off = 0.0 if not offset else (offset / 2 ** 16 if isinstance(offset, int) else float(offset))
    lo = 0.0 if not clip else (clip / 2 ** 16 if isinstance(clip, int) else float(clip))
    m = None if mask is None else torch.from_numpy(np.ascontiguousarray(np.asarray(mask).astype(np.uint8)))
    return hp.preprocess_u16(raw, m, off, lo, bool(log_correct), float(scale))
