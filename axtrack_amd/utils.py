"""The one function of the reference's utils.py that sits on the hot path's boundary: `_compute_astar_path`
(axtrack/utils.py:351-390), re-exported by the reference's package (__init__.py:16)."""
import numpy as np
import torch

from . import hotpath as hp


def _compute_astar_path(source, target, weights, return_dist=True, max_path_length=10000, device='cuda:0', conn8=False):
    """utils.py:351-390: the A* path between two (y, x) coordinates on a weight matrix, as a bool scipy.sparse.coo_matrix of
    the weight matrix's shape marking the path's cells (both end points included), or None if the path has more than
    max_path_length cells; with return_dist also its length in cells ((None, None) without a path).

    The reference hands this to pyastar2d (a C++ extension that is absent from its tree); here the search runs on the GPU
    (axt_path_cells). `weights` must be what the reference passes (AxonDetections.py:598): 1 on the mask, 65536 off it (f32
    [H, W]) -- or all ones, where the answer is a closed-form staircase (columns first, then rows; which of the equally
    short paths pyastar2d returns is not pinned). An end point outside the grid has no path."""
    from scipy import sparse
    w = np.asarray(weights)
    if w.ndim != 2:
        raise ValueError(f'weights must be [H, W], got {w.shape}')
    H, W = w.shape
    on = w == 1
    if not np.all(on | (w == 2 ** 16)):
        raise ValueError("weights other than the reference's mask weights {1 on the mask, 65536 off it} are not supported")
    (ya, xa), (yb, xb) = (int(source[0]), int(source[1])), (int(target[0]), int(target[1]))
    none = (None, None) if return_dist else None
    if not (0 <= ya < H and 0 <= xa < W and 0 <= yb < H and 0 <= xb < W):
        return none
    if on.all():
        sx, sy = (1 if xb >= xa else -1), (1 if yb >= ya else -1)
        if conn8:
            k = min(abs(xb - xa), abs(yb - ya))
            dx, dy = abs(xb - xa) - k, abs(yb - ya) - k
            c = np.concatenate([xa + sx * np.arange(k + 1), xa + sx * (k + np.arange(1, dx + 1)), np.full(dy, xb)])
            r = np.concatenate([ya + sy * np.arange(k + 1), np.full(dx, ya + sy * k), ya + sy * (k + np.arange(1, dy + 1))])
        else:
            xs, ys = np.arange(xa, xb + sx, sx), np.arange(ya, yb + sy, sy)
            r = np.concatenate([np.full(len(xs), ya), ys[1:]])
            c = np.concatenate([xs, np.full(len(ys) - 1, xb)])
        if len(r) > max_path_length:
            return none
    else:
        limit = int(min(max_path_length + 1, 32767))          # axt_path_cells reports paths of fewer than `limit` cells
        dev = torch.device(device)
        t = lambda v: torch.tensor([v], dtype=torch.int32, device=dev)
        D, cells = hp.path_cells(t(xa), t(ya), t(xb), t(yb), H, W, on, limit, conn8)
        n = int(D[0, 0].item())
        if n >= limit:
            return none
        cell = cells[0, 0, :n].cpu().numpy().astype(np.int64)
        r, c = cell // W, cell % W
    path = sparse.coo_matrix((np.ones(len(r)), (r.astype(np.int64), c.astype(np.int64))), (H, W), bool)
    return (path, len(r)) if return_dist else path
