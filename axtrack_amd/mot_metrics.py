"""Multi-object-tracking scores of an association against labelled identities -- what the reference's
hyper-parameter search reads per parameter combination (AxonDetections.py:880-895):

    acc = mm.utils.compare_to_groundtruth(target, pred, dist='euclidean', distth=nms_min_dist**2)
    res = mm.metrics.create().compute(acc, metrics=mm.metrics.motchallenge_metrics)

py-motmetrics (pinned `motmetrics==1.1.3`, axtr.yml:127) is not part of the reference tree and not installed
here, so this is a restatement of its published algorithm -- the CLEAR-MOT event accumulator (Bernardin &
Stiefelhagen 2008: keep last frame's correspondences while they stay within the threshold, assign the rest at minimum
total distance, a changed partner is a SWITCH) and the identity scores of Ristani et al. 2016 (one global bipartite
matching of label identities to predicted identities) -- and its numbers are NOT pinned against it. Where several
assignments tie (equal summed distances) the library's choice depends on its LAP backend; scipy's is used here.

Host code: evaluation of finished associations (tens of small assignment problems), not part of the inference path.
"""
import numpy as np
import pandas as pd
from scipy.optimize import linear_sum_assignment

# mm.metrics.motchallenge_metrics of motmetrics 1.1.3, in its order
MOTCHALLENGE_METRICS = ['idf1', 'idp', 'idr', 'recall', 'precision', 'num_unique_objects', 'mostly_tracked',
                        'partially_tracked', 'mostly_lost', 'num_false_positives', 'num_misses', 'num_switches',
                        'num_fragmentations', 'mota', 'motp']


def _assign(cost):
    """Minimum-cost assignment where NaN marks pairs that may not be matched: as many admissible pairs as possible
    first, then the smallest sum (the library pads NaN with a constant larger than any admissible total)."""
    ok = np.isfinite(cost)
    if not ok.any():
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    big = (min(cost.shape) + 1) * (np.abs(cost[ok]).max() + 1.0)
    r, c = linear_sum_assignment(np.where(ok, cost, big))
    keep = ok[r, c]
    return r[keep], c[keep]


def sq_distances(obj_xy, hyp_xy, max_d2):
    """norm2squared_matrix: squared euclidean distances, NaN beyond max_d2."""
    o = np.asarray(obj_xy, np.float64).reshape(-1, 2)
    h = np.asarray(hyp_xy, np.float64).reshape(-1, 2)
    d = ((o[:, None, :] - h[None, :, :]) ** 2).sum(-1)
    d[d > max_d2] = np.nan
    return d


def accumulate(frames, max_d2):
    """frames: per frame (obj_ids, obj_xy [no,2], hyp_ids, hyp_xy [nh,2]). Returns the event log as a dict of
    arrays: per event frame, kind ('MATCH', 'SWITCH', 'MISS', 'FP'), object id, hypothesis id (-1 = none),
    distance (NaN = none); plus the raw co-occurrences (object, hypothesis) within max_d2 and the number of frames."""
    ev_f, ev_k, ev_o, ev_h, ev_d = [], [], [], [], []
    raw_o, raw_h = [], []
    partner = {}                                     # object id -> hypothesis id it was last matched with
    for f, (oids, oxy, hids, hxy) in enumerate(frames):
        oids = np.asarray(oids, np.int64); hids = np.asarray(hids, np.int64)
        no, nh = len(oids), len(hids)
        o_done = np.zeros(no, bool); h_done = np.zeros(nh, bool)
        if no and nh:
            d = sq_distances(oxy, hxy, max_d2)
            ri, rj = np.nonzero(np.isfinite(d))
            raw_o += oids[ri].tolist(); raw_h += hids[rj].tolist()
            # 1. correspondences of earlier frames survive while they stay admissible
            for i in range(no):
                hp = partner.get(int(oids[i]))
                if hp is None:
                    continue
                j = np.nonzero(~h_done & (hids == hp))[0]
                if len(j) and np.isfinite(d[i, j[0]]):
                    j = j[0]
                    o_done[i] = h_done[j] = True
                    ev_f.append(f); ev_k.append('MATCH'); ev_o.append(int(oids[i])); ev_h.append(int(hids[j])); ev_d.append(d[i, j])
            # 2. the rest at minimum total distance
            d2 = d.copy()
            d2[o_done, :] = np.nan
            d2[:, h_done] = np.nan
            for i, j in zip(*_assign(d2)):
                o, h = int(oids[i]), int(hids[j])
                switch = o in partner and partner[o] != h
                ev_f.append(f); ev_k.append('SWITCH' if switch else 'MATCH'); ev_o.append(o); ev_h.append(h); ev_d.append(d[i, j])
                o_done[i] = h_done[j] = True
                partner[o] = h
        for o in oids[~o_done]:
            ev_f.append(f); ev_k.append('MISS'); ev_o.append(int(o)); ev_h.append(-1); ev_d.append(np.nan)
        for h in hids[~h_done]:
            ev_f.append(f); ev_k.append('FP'); ev_o.append(-1); ev_h.append(int(h)); ev_d.append(np.nan)
    return dict(frame=np.array(ev_f, np.int64), kind=np.array(ev_k, dtype=object), obj=np.array(ev_o, np.int64),
                hyp=np.array(ev_h, np.int64), dist=np.array(ev_d, np.float64),
                raw_obj=np.array(raw_o, np.int64), raw_hyp=np.array(raw_h, np.int64), n_frames=len(frames),
                obj_seen=np.concatenate([np.asarray(fr[0], np.int64) for fr in frames] + [np.zeros(0, np.int64)]),
                hyp_seen=np.concatenate([np.asarray(fr[2], np.int64) for fr in frames] + [np.zeros(0, np.int64)]))


def summarize(ev):
    """The fifteen MOTChallenge scores of an event log (pd.Series, index MOTCHALLENGE_METRICS)."""
    kind, obj = ev['kind'], ev['obj']
    n_match = int((kind == 'MATCH').sum()); n_switch = int((kind == 'SWITCH').sum())
    n_miss = int((kind == 'MISS').sum()); n_fp = int((kind == 'FP').sum())
    n_det = n_match + n_switch
    objects, freq = np.unique(obj[obj >= 0], return_counts=True)
    n_obj = int(freq.sum())
    tracked = np.array([((obj == o) & (kind != 'MISS')).sum() for o in objects], np.float64)
    ratio = tracked / np.maximum(freq, 1)
    frag = 0
    for o in objects:
        missed = (kind[obj == o] == 'MISS').astype(np.int64)              # the object's events, in frame order
        seen = np.nonzero(missed == 0)[0]
        if len(seen):
            frag += int((np.diff(missed[seen[0]:seen[-1] + 1]) == 1).sum())
    # identity scores: one global matching of label identities to predicted identities
    o_ids, o_cnt = np.unique(ev['obj_seen'], return_counts=True)
    h_ids, h_cnt = np.unique(ev['hyp_seen'], return_counts=True)
    no, nh = len(o_ids), len(h_ids)
    n_pred = int(h_cnt.sum())
    fp = np.zeros((no + nh, no + nh)); fn = np.zeros((no + nh, no + nh))
    fp[no:, :nh] = np.nan; fn[:no, nh:] = np.nan
    for r in range(no):
        fn[r, :nh] = o_cnt[r]; fn[r, nh + r] = o_cnt[r]
    for c in range(nh):
        fp[:no, c] = h_cnt[c]; fp[c + no, c] = h_cnt[c]
    if len(ev['raw_obj']):
        r = np.searchsorted(o_ids, ev['raw_obj']); c = np.searchsorted(h_ids, ev['raw_hyp'])
        np.subtract.at(fp, (r, c), 1.0); np.subtract.at(fn, (r, c), 1.0)
    rr, cc = _assign(fp + fn)
    idfp, idfn = float(fp[rr, cc].sum()), float(fn[rr, cc].sum())
    idtp = n_obj - idfn

    def div(a, b):
        return a / b if b else np.nan
    vals = dict(idf1=div(2 * idtp, n_obj + n_pred), idp=div(idtp, idtp + idfp), idr=div(idtp, idtp + idfn),
                recall=div(n_det, n_obj), precision=div(n_det, n_fp + n_det), num_unique_objects=len(objects),
                mostly_tracked=int((ratio >= 0.8).sum()), partially_tracked=int(((ratio >= 0.2) & (ratio < 0.8)).sum()),
                mostly_lost=int((ratio < 0.2).sum()), num_false_positives=n_fp, num_misses=n_miss, num_switches=n_switch,
                num_fragmentations=frag, mota=1.0 - div(n_miss + n_switch + n_fp, n_obj) if n_obj else np.nan,
                motp=div(float(np.nansum(ev['dist'])), n_det))
    return pd.Series([vals[k] for k in MOTCHALLENGE_METRICS], index=MOTCHALLENGE_METRICS)


def compare_to_groundtruth(target, pred, max_d2):
    """The two libmot tables (index (FrameId, Id), columns X, Y, ... as AxonDetections.det2libmot_det makes them) ->
    event log over the union of their frames."""
    def per_frame(tab):
        if tab is None or not len(tab):
            return {}
        fr = tab.index.get_level_values('FrameId').to_numpy().astype(np.int64)
        ids = tab.index.get_level_values('Id').to_numpy().astype(np.int64)
        xy = tab[['X', 'Y']].to_numpy(np.float64)
        return {f: (ids[fr == f], xy[fr == f]) for f in np.unique(fr)}
    t, p = per_frame(target), per_frame(pred)
    empty = (np.zeros(0, np.int64), np.zeros((0, 2)))
    frames = [t.get(f, empty) + p.get(f, empty) for f in sorted(set(t) | set(p))]
    return accumulate(frames, max_d2)
