"""The CPU oracle against the golden vectors captured from the reference's own Python
(tests/golden/make_golden.py). Pins oracle/ for SURVEY.md rows a-1 ... a-8, a-10, a-11, a-13."""
import numpy as np
import pytest

from axtrack_amd import synth
from oracle import oracle as orc
from helpers import golden_dets

# CNN parity is by tolerance: the reference's conv/linear run through oneDNN/MKL with an
# unspecified f32 summation order (model.py:50-53).
CNN_ATOL, CNN_RTOL = 1e-5, 1e-5


def split(counts, *arrs):
    offs = np.concatenate([[0], np.cumsum(counts)])
    return [[a[offs[i]:offs[i + 1]] for a in arrs] for i in range(len(counts))]


def test_cnn_forward_matches_reference(golden, weights):
    g = golden('cnn_512')
    frames = synth.synth_frames(int(g['T_all']), int(g['H']), int(g['W']), seed=int(g['frames_seed']))
    X = np.stack([frames[t:t + 5] for t in range(2)])
    y = orc.cnn_forward(weights, X)
    np.testing.assert_allclose(y, g['yolo'][:2], atol=CNN_ATOL, rtol=CNN_RTOL)


@pytest.mark.parametrize('name', ['detect_1024', 'detect_ragged'])
def test_tiling_keeps_the_reference_tiles(golden, name):
    g = golden(name)
    frames = synth.synth_frames(int(g['T_all']), int(g['H']), int(g['W']), seed=int(g['frames_seed']))
    zt = g['zero_tile']
    if zt[0] >= 0:
        frames[:, zt[0] * 512:(zt[0] + 1) * 512, zt[1] * 512:(zt[1] + 1) * 512] = 0
    keep = orc.kept_tiles(frames)
    ref_keep = [tuple(ix) for ix in np.argwhere(g['kept_tiles'])]
    assert keep == ref_keep
    # the tile stack of frame 0 reproduces the reference's YOLO output for that frame
    if name == 'detect_ragged':
        from axtrack_amd import synth as s
        y = orc.cnn_forward(s.synth_state_dict(42), orc.frame_tile_stack(frames, 0, keep)[:2])
        np.testing.assert_allclose(y, g['yolo'][0, :2], atol=CNN_ATOL, rtol=CNN_RTOL)


@pytest.mark.parametrize('name', ['detect_1024', 'detect_ragged', 'detect_crafted'])
def test_decode_stitch_nms_bit_exact(golden, name):
    """Given the reference's YOLO tensors, decode/filter/stitch/NMS must reproduce the
    reference's detection tables exactly (conf bits, integer anchors, order)."""
    g = golden(name)
    if name == 'detect_crafted':
        keep = [(0, 0), (0, 1), (1, 0), (1, 1)]
    else:
        keep = [tuple(ix) for ix in np.argwhere(g['kept_tiles'])]
    dets = orc.detect_from_yolo(list(g['yolo']), keep)
    ref = split(g['counts'], g['conf'], g['x'], g['y'])
    for t, ((c, x, y), (rc, rx, ry)) in enumerate(zip(dets, ref)):
        if name == 'detect_crafted' and t == 1:
            continue        # exact confidence ties: order is sort-implementation specific, see below
        assert len(c) == len(rc), f'frame {t}'
        assert np.array_equal(c.view(np.uint32), rc.view(np.uint32)), f'frame {t}'
        assert np.array_equal(x, rx) and np.array_equal(y, ry), f'frame {t}'


def test_nms_with_confidence_ties_is_a_valid_greedy_result(golden):
    """Frame 1 of the crafted fixture has hundreds of exactly tied confidences. The reference
    sorts with pandas' default (unstable) quicksort, so WHICH of two tied neighbours survives
    is an implementation detail; the oracle defines ties by (tile, cell) order. Both must be
    valid greedy results: survivors pairwise >= 23 px apart, every candidate kept or within
    23 px of a survivor whose confidence is >= its own."""
    g = golden('detect_crafted')
    keep = [(0, 0), (0, 1), (1, 0), (1, 1)]
    cand = orc.stitch(orc.decode_filter(g['yolo'][1]), keep)
    ref = split(g['counts'], g['conf'], g['x'], g['y'])[1]
    mine = orc.nms(*cand)
    for c, x, y in (ref, mine):
        d2 = (x[:, None] - x[None]) ** 2 + (y[:, None] - y[None]) ** 2
        np.fill_diagonal(d2, 10 ** 9)
        assert d2.min() >= 529
        cd2 = (cand[1][:, None] - x[None]) ** 2 + (cand[2][:, None] - y[None]) ** 2
        covered = ((cd2 < 529) & (c[None] >= cand[0][:, None])).any(1) | (cd2 == 0).any(1)
        assert covered.all()
        assert np.all(np.diff(c.astype(np.float64)) <= 0)
    assert abs(len(mine[0]) - len(ref[0])) <= 12


def test_tiled_tables_before_stitching(golden):
    g = golden('detect_1024')
    tiled = g['tiled']
    for t in range(g['yolo'].shape[0]):
        per_tile = orc.decode_filter(g['yolo'][t])
        for k, (c, x, y, cell) in enumerate(per_tile):
            ref = tiled[(tiled[:, 0] == t) & (tiled[:, 1] == k)]
            order = np.lexsort((cell, c))
            assert np.array_equal(c[order].astype(np.float64), ref[:, 2])
            assert np.array_equal(x[order], ref[:, 3].astype(np.int64))
            assert np.array_equal(y[order], ref[:, 4].astype(np.int64))


def test_libmot_format_capping_and_costs(golden):
    g, a = golden('detect_1024'), golden('assoc_parts')
    dets = split(g['counts'], g['conf'], g['x'], g['y'])
    rows = orc.libmot_rows(dets)
    assert np.array_equal(rows, a['libmot'])
    assert np.array_equal(orc.cap_conf(rows[:, 6], 'scale_to_max'), a['conf_scale_to_max'])
    assert np.array_equal(orc.cap_conf(rows[:, 6], 'ceil'), a['conf_ceil'])
    obs = orc.observation_cost(orc.cap_conf(rows[:, 6], 'scale_to_max'))
    assert np.array_equal(obs, a['obs_cost'])
    D = np.arange(1, 501)
    for gap in (1, 2):
        assert np.array_equal(orc.transition_cost(D, gap), a[f'trans_cost_gap{gap}'])
    # edge admission thresholds that follow
    assert (orc.transition_cost(D, 1) < 0.7).sum() == 251
    assert (orc.transition_cost(D, 2) < 0.7).sum() == 86


def test_ided_dets_all_with_empty_frame_quirk(golden):
    g, a = golden('detect_1024'), golden('assoc_parts')
    dets = split(g['counts'], g['conf'], g['x'], g['y'])
    trajs = {}
    for tid, f, k in a['traj']:
        trajs.setdefault(int(tid), []).append((int(f), int(k)))
    trajs = [trajs[i] for i in sorted(trajs)]
    tables = orc.ided_tables(trajs, dets)
    assert [len(p) for p in tables] == list(a['ided_frame_counts'])
    ids, labels, info, vals = orc.ided_dets_all(tables)
    assert [f'Axon_{i:0>3}' for i in ids] == list(a['ided_all_index'])
    assert np.array_equal(labels, a['ided_all_cols_frame'])
    assert list(info) == list(a['ided_all_cols_info'])
    np.testing.assert_array_equal(vals, a['ided_all_values'])


def test_appearance_features_restatement_is_self_consistent():
    """f-3: the C restatement of calcHist/normalize/compareHist against an independent numpy formulation of the same
    published behaviour (cv2 is absent: this pins the oracle to its own specification, not to OpenCV)."""
    rng = np.random.default_rng(12)
    img = (rng.random((120, 150)) * 1.25).astype(np.float32)
    img[rng.random(img.shape) < 0.5] = 0
    x = np.array([0, 149, 75, -20, 200, 35]); y = np.array([0, 119, 60, 130, 50, 35])
    h = orc.box_histograms(img, x, y)
    for k in range(len(x)):
        r0, c0 = max(int(y[k]) - 35, 0), max(int(x[k]) - 35, 0)
        crop = img[r0:r0 + 70, c0:c0 + 70]
        idx = np.floor(crop.astype(np.float64) * 180.0).astype(np.int64)
        cnt = np.bincount(idx[(idx >= 0) & (idx < 180)], minlength=180).astype(np.float32)
        rg = float(cnt.max()) - float(cnt.min())
        sc = 1.0 / rg if rg > np.finfo(np.float64).eps else 0.0
        ref = cnt * np.float32(sc) + np.float32(-float(cnt.min()) * sc)
        assert np.array_equal(h[k], ref), k
    assert not h[4].any()                                   # box entirely right of the image: empty crop
    d = orc.bhattacharyya(h, h)
    assert np.all(np.diag(d)[[0, 1, 2, 3, 5]] < 1e-7) and np.allclose(d, d.T, atol=1e-15)
    a, b = h[0].astype(np.float64), h[2].astype(np.float64)
    ref = np.sqrt(max(1 - np.sqrt(a * b).sum() / np.sqrt(a.sum() * b.sum()), 0))
    assert abs(d[0, 2] - ref) < 1e-12
    # the transition cost with the term: monotone in the similarity, equal to the table model at weight 0
    D = np.array([[10, 200, 400]])
    c0 = orc.transition_cost(D, 1)
    c1 = orc.transition_cost(D, 1, vis_w=0.3, vis_sim=np.array([[1.0, 1.0, 1.0]]))
    c2 = orc.transition_cost(D, 1, vis_w=0.3, vis_sim=np.array([[0.0, 0.0, 0.0]]))
    assert np.all(c1 < c2) and np.all(c1[0, 1:] < c0[0, 1:])


def test_detection_metrics_match_the_reference(golden):
    """f-4: TP/FP/FN over the 13 confidence thresholds and precision/recall/F1, against what the reference's own
    compute_TP_FP_FN / compute_prc_rcl_F1 returned for the detect_1024 detections and synthetic labels."""
    g, m = golden('detect_1024'), golden('metrics')
    dets = golden_dets(g)
    assert np.array_equal(orc.all_conf_thrs(), m['all_conf_thrs']) and int(m['nms_min_dist']) == 23
    offs = np.concatenate([[0], np.cumsum(m['gt_counts'])])
    doffs = np.concatenate([[0], np.cumsum(g['counts'])])
    k_bbox = int(np.where(m['all_conf_thrs'] == 0.7)[0][0])
    for t, det in enumerate(dets):
        gx, gy = m['gt_x'][offs[t]:offs[t + 1]], m['gt_y'][offs[t]:offs[t + 1]]
        cm = orc.detection_confusion(det, gx, gy)
        assert np.array_equal(cm, m['confusion'][t]), t
        assert np.array_equal(orc.prc_rcl_f1(cm), m['prc_rcl_f1'][t])
        fp, fn = orc.detection_confusion(det, gx, gy, return_masks_at=k_bbox)
        assert np.array_equal(fp, m['fp_mask_at_bbox_thr'][doffs[t]:doffs[t + 1]])
        n_gt = max(len(gx), 1)
        lo = offs[t] + sum(1 for q in range(t) if m['gt_counts'][q] == 0)      # an empty frame contributes one phantom row
        assert np.array_equal(fn, m['fn_mask_at_bbox_thr'][lo:lo + n_gt])
