"""Known-answer cases for the tracking scores of the hyper-parameter search (axtrack_amd/mot_metrics.py). The
library the reference calls (motmetrics 1.1.3) is absent here: these cases are worked by hand from the published
definitions, they do not pin the restatement against the library."""
import numpy as np
import pandas as pd
import pytest

from axtrack_amd import mot_metrics as mm


def frames_from(gt, hyp, n):
    """gt / hyp: {id: {frame: (x, y)}} -> accumulate()'s input."""
    out = []
    for f in range(n):
        o = [(i, p[f]) for i, p in gt.items() if f in p]
        h = [(i, p[f]) for i, p in hyp.items() if f in p]
        out.append(([i for i, _ in o], np.array([p for _, p in o], float).reshape(-1, 2),
                    [i for i, _ in h], np.array([p for _, p in h], float).reshape(-1, 2)))
    return out


def score(gt, hyp, n, max_d2=529.0):
    return mm.summarize(mm.accumulate(frames_from(gt, hyp, n), max_d2))


def test_perfect_tracking():
    gt = {1: {f: (10 * f, 5) for f in range(5)}, 2: {f: (300, 10 * f) for f in range(5)}}
    s = score(gt, {7: gt[1], 8: gt[2]}, 5)
    assert list(s.index) == mm.MOTCHALLENGE_METRICS
    assert s.mota == 1 and s.idf1 == 1 and s.idp == 1 and s.idr == 1 and s.motp == 0
    assert s.recall == 1 and s.precision == 1
    assert (s.num_unique_objects, s.mostly_tracked, s.partially_tracked, s.mostly_lost) == (2, 2, 0, 0)
    assert (s.num_false_positives, s.num_misses, s.num_switches, s.num_fragmentations) == (0, 0, 0, 0)


def test_identity_switch():
    gt = {1: {f: (10 * f, 0) for f in range(5)}}
    hyp = {10: {f: (10 * f, 0) for f in range(3)}, 11: {f: (10 * f, 0) for f in (3, 4)}}
    s = score(gt, hyp, 5)
    assert s.num_switches == 1 and s.num_misses == 0 and s.num_false_positives == 0
    assert s.mota == pytest.approx(0.8)
    assert (s.idp, s.idr, s.idf1) == (pytest.approx(0.6), pytest.approx(0.6), pytest.approx(0.6))


def test_miss_false_positive_and_fragmentation():
    gt = {1: {f: (10 * f, 0) for f in range(5)}}
    hyp = {10: {f: (10 * f, 0) for f in (0, 1, 3, 4)}, 99: {2: (400, 400)}}
    s = score(gt, hyp, 5)
    assert (s.num_misses, s.num_false_positives, s.num_fragmentations, s.num_switches) == (1, 1, 1, 0)
    assert s.mota == pytest.approx(0.6) and s.recall == pytest.approx(0.8) and s.precision == pytest.approx(0.8)
    assert s.mostly_tracked == 1
    assert s.idf1 == pytest.approx(0.8)


def test_correspondences_persist_while_admissible():
    """CLEAR-MOT keeps last frame's pairs although swapping them would be cheaper."""
    gt = {1: {0: (0, 0), 1: (0, 0)}, 2: {0: (10, 0), 1: (10, 0)}}
    hyp = {5: {0: (0, 0), 1: (6, 0)}, 6: {0: (10, 0), 1: (4, 0)}}
    s = score(gt, hyp, 2)
    assert s.num_switches == 0 and s.motp == pytest.approx(18.0)
    # without the first frame the cheaper pairing is taken
    s = mm.summarize(mm.accumulate(frames_from(gt, hyp, 2)[1:], 529.0))
    assert s.motp == pytest.approx(16.0)


def test_distance_threshold_is_inclusive():
    gt = {1: {0: (0, 0)}}
    assert score(gt, {9: {0: (23, 0)}}, 1).num_misses == 0                    # 529 <= 529
    s = score(gt, {9: {0: (23, 1)}}, 1)                                       # 530
    assert (s.num_misses, s.num_false_positives) == (1, 1) and s.mota == pytest.approx(-1.0)


def test_never_tracked_object_is_mostly_lost_and_empty_prediction():
    gt = {1: {f: (0, 0) for f in range(4)}}
    s = score(gt, {}, 4)
    assert s.mostly_lost == 1 and s.num_misses == 4 and s.mota == 0 and s.idf1 == 0 and s.recall == 0
    assert np.isnan(s.precision) and np.isnan(s.motp)


def test_libmot_tables_as_input():
    """compare_to_groundtruth takes the two (FrameId, Id)-indexed tables of det2libmot_det."""
    def table(rows):
        return pd.DataFrame(rows, columns=['FrameId', 'Id', 'X', 'Y', 'Width', 'Height', 'conf']).set_index(['FrameId', 'Id'])
    target = table([[0, 1, 0, 0, 70, 70, 1], [1, 1, 5, 0, 70, 70, 1], [2, 1, 9, 0, 70, 70, 1]])
    pred = table([[0, 4, 1, 0, 70, 70, .9], [2, 4, 9, 3, 70, 70, .8], [3, 4, 9, 9, 70, 70, .8]])
    s = mm.summarize(mm.compare_to_groundtruth(target, pred, 529.0))
    assert (s.num_misses, s.num_false_positives, s.num_switches, s.num_fragmentations) == (1, 1, 0, 1)
    assert s.motp == pytest.approx((1 + 9) / 2)
    assert mm.summarize(mm.compare_to_groundtruth(target, None, 529.0)).num_misses == 3
