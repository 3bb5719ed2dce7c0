// Global data association as successive shortest paths by dynamic-programming sweeps over the frames -- host
// statement of the algorithm that mcf_gpu.hip runs on the device (same phases, same tie rules, same results).
//
// Replaces libmot.data_association.MinCostFlowTracker.compute_trajectories() as driven by the reference at
// axtrack/AxonDetections.py:663-690 (libmot is absent from the reference tree; network and stop rule as in mcf.cpp).
//
// The tracking network is a DAG layered by frames (arcs reach at most `max_gap` frames ahead), and the optimum has few
// unit flows (tracks) compared with detections. Successive shortest paths needs one shortest S->T path per track; on
// this network a shortest-path computation is a sweep over the frames in reverse time,
//     hV[k] = min(exit_k, min over out-arcs (cost + hU[head]))          distance of v_k to T
//     hU[k] = obs_k + hV[k]                                              (k unused)
// a pure data-parallel pass per frame with no priority queue -- which is what maps onto the GPU. Arcs that carry flow
// turn around in the residual network and point backwards in time:
//     u_b -> v_a  for the flow-carrying transition a->b   (cost -c_ab):   hU[b] = -c_ab + hV[a]   "steal b from a"
//     v_k -> u_k  for a used detection                     (cost -obs_k):  hV[k] = min(., -obs_k + hU[k]) "drop k"
// The first kind is resolved inside the step of a's frame by iterating that step to a fixed point (a's alternatives
// may in turn steal), the second by a pass along the tracks after the sweep; what a sweep cannot see -- a value lowered
// after an earlier step of the same sweep has already read it -- is detected exactly (no arc may remain that can still
// be relaxed) and answered with another sweep that starts from the labels reached so far. Labels only ever decrease and each is the length of a walk to T,
// so the sweeps end at the exact distances (the residual network of a min-cost flow has no negative cycle).
// Stop rule as in mcf.cpp: push while F < min_flow, or while F < max_flow and the next path is negative.
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>

#include <vector>

#include "../../include/axtrack_hip.h"

void axt_set_error(const char *fmt, ...);

namespace {

constexpr int64_t INF = INT64_MAX / 4;
constexpr int32_t NONE = -1, TERMINAL = -2, REVOBS = -3;   // succ/pred: TERMINAL = exit to T / entry from S

struct Dag {
    int n, F;
    const int64_t *fp;                    // frame_ptr [F+1]
    const int64_t *obs, *entry, *exitc, *row_ptr, *cost;
    const int32_t *col;
    std::vector<int32_t> frame_of;
    std::vector<int32_t> succ, pred_arc, pred_tail;
    std::vector<int64_t> hU, hV, Bb;
    std::vector<int32_t> bestV;
    size_t stat_sweeps = 0, stat_steps = 0, stat_iters = 0;

    bool used(int k) const { return pred_arc[k] != NONE; }

    // one step of the sweep: frame t, iterated until the steals out of this frame are stable
    bool step(int t, int max_gap, bool &dirty)
    {
        const int lo = (int)fp[t], hi = (int)fp[t + 1];
        const int wend = (int)fp[t + 1 + max_gap < F ? t + 1 + max_gap : F];
        for (int it = 0;; ++it) {
            ++stat_iters;
            for (int j = lo; j < hi; ++j) {                                   // phase A
                int64_t best = INF;
                int32_t bv = NONE;
                if (succ[j] != TERMINAL) { best = exitc[j]; bv = TERMINAL; }
                for (int64_t e = row_ptr[j]; e < row_ptr[j + 1]; ++e) {
                    if ((int32_t)e == succ[j]) continue;
                    const int64_t hu = hU[col[e]];
                    if (hu >= INF) continue;
                    const int64_t cand = cost[e] + hu;
                    if (cand < best) { best = cand; bv = (int32_t)e; }
                }
                if (used(j) && hU[j] < INF) {
                    const int64_t cand = -obs[j] + hU[j];
                    if (cand < best) { best = cand; bv = REVOBS; }
                }
                hV[j] = best;
                bestV[j] = bv;
                if (!used(j)) hU[j] = best < INF ? obs[j] + best : INF;
            }
            bool changed = false;                                             // phase B
            for (int b = hi; b < wend; ++b) {
                if (pred_arc[b] < 0) continue;
                const int a = pred_tail[b];
                if (a < lo || a >= hi || hV[a] >= INF) continue;
                const int64_t nv = -cost[pred_arc[b]] + hV[a];
                if (nv < hU[b]) {
                    hU[b] = nv;
                    changed = true;
                    if (nv < Bb[b]) dirty = true;        // a tail between the two frames read the older value and would have used this one
                }
            }
            if (!changed) break;
            if (it > 4 * (hi - lo) + 16) return false;
        }
        // tails of this frame that read hU[b] of a used b: remember what value of hU[b] would have changed their choice
        // (hU[b] may still be lowered in this sweep, at the step of b's predecessor or by the pass along the tracks)
        for (int j = lo; j < hi; ++j)
            for (int64_t e = row_ptr[j]; e < row_ptr[j + 1]; ++e) {
                const int b = col[e];
                if (pred_arc[b] < 0 || (int32_t)e == succ[j]) continue;
                const int64_t v = hV[j] >= INF ? INF : hV[j] - cost[e];
                if (v > Bb[b]) Bb[b] = v;
            }
        return true;
    }

    // exact distances to T in the residual network; false on a (theoretically impossible) failure to converge
    bool distances(int max_gap)
    {
        for (int k = 0; k < n; ++k) { hU[k] = INF; hV[k] = INF; bestV[k] = NONE; }
        for (int sweep = 0;; ++sweep) {
            ++stat_sweeps;
            bool dirty = false;
            for (int k = 0; k < n; ++k) Bb[k] = -INF;
            for (int t = F - 1; t >= 0; --t) {
                ++stat_steps;
                if (!step(t, max_gap, dirty)) return false;
            }
            // forward pass along the tracks (cheap: used detections only): a walk that runs backwards along a track for
            // several frames -- steal c, drop its predecessor k, drop k's predecessor ... -- is propagated in one pass;
            // a value that a reader of the finished sweep would have used calls for another sweep
            for (int k = 0; k < n; ++k) {
                if (pred_arc[k] < 0) continue;
                const int a = pred_tail[k];
                if (hV[a] < INF) {
                    const int64_t nv = -cost[pred_arc[k]] + hV[a];
                    if (nv < hU[k]) { hU[k] = nv; if (nv < Bb[k]) dirty = true; }
                }
                if (hU[k] < INF && succ[k] >= 0 && -obs[k] + hU[k] < hV[k]) { hV[k] = -obs[k] + hU[k]; bestV[k] = REVOBS; }
            }
            if (!dirty) return true;
            if (sweep > 2 * F + 16) return false;
        }
    }

    // walk the shortest path from S and flip its arcs; false if the labels do not describe a simple path
    bool augment(int k0)
    {
        struct Hop { int32_t kind, a, b, arc; };     // kind 0: entry->u_a | 1: u_a->v_a | 2: v_a->u_b (arc) | 3: v_a->T
        std::vector<Hop> add;                         //      4: u_b->v_a reverse of arc | 5: v_a->u_a reverse obs
        int k = k0;
        bool at_u = true, via_forward = true;         // how the walk arrived at u_k
        add.push_back(Hop{0, k0, -1, -1});
        for (int hops = 0; hops < 4 * n + 8; ++hops) {
            if (at_u) {
                if (!used(k)) { add.push_back(Hop{1, k, -1, -1}); at_u = false; via_forward = true; continue; }
                if (pred_arc[k] < 0) return false;    // a track start cannot be stolen or dropped through S
                add.push_back(Hop{4, pred_tail[k], k, pred_arc[k]});
                k = pred_tail[k];
                at_u = false;
                via_forward = false;
            } else {
                const int32_t bv = bestV[k];
                if (bv == TERMINAL) { add.push_back(Hop{3, k, -1, -1}); goto apply; }
                if (bv == REVOBS) {
                    if (via_forward) return false;
                    add.push_back(Hop{5, k, -1, -1});
                    at_u = true;                      // arrives at u_k through the reversed observation arc: k is dropped,
                    // so the next hop must release its predecessor; used(k) still holds (state is applied afterwards)
                    continue;
                }
                if (bv < 0) return false;
                add.push_back(Hop{2, k, col[bv], bv});
                k = col[bv];
                at_u = true;
                via_forward = true;
            }
        }
        return false;
    apply:
        for (const Hop &h : add) {                    // removals first
            if (h.kind == 4) { succ[h.a] = NONE; pred_arc[h.b] = NONE; pred_tail[h.b] = -1; }
            if (h.kind == 5) { /* k unused: its predecessor link is removed by the kind-4 hop that follows */ }
        }
        for (const Hop &h : add) {
            if (h.kind == 0) { pred_arc[h.a] = TERMINAL; pred_tail[h.a] = -1; }
            else if (h.kind == 2) { succ[h.a] = h.arc; pred_arc[h.b] = h.arc; pred_tail[h.b] = h.a; }
            else if (h.kind == 3) succ[h.a] = TERMINAL;
        }
        for (const Hop &h : add)
            if (h.kind == 5) { pred_arc[h.a] = NONE; pred_tail[h.a] = -1; succ[h.a] = NONE; }
        return true;
    }
};

}  // namespace

extern "C" int axt_mcf_solve_dag(int n_det, int n_frames, const int64_t *h_frame_ptr, const int64_t *h_obs,
                                 const int64_t *h_entry, const int64_t *h_exit, const int64_t *h_row_ptr,
                                 const int32_t *h_col, const int64_t *h_cost, int min_flow, int max_flow,
                                 int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost)
{
    if (n_det < 0 || n_frames < 0 || !h_frame_ptr || !h_row_ptr || !n_tracks || !total_cost ||
        (n_det > 0 && (!h_obs || !h_entry || !h_exit || !h_next || !h_track))) {
        axt_set_error("axt_mcf_solve_dag: null or negative argument");
        return AXT_EINVAL;
    }
    if (min_flow < 0 || max_flow < min_flow) {
        axt_set_error("axt_mcf_solve_dag: need 0 <= min_flow <= max_flow (got %d, %d)", min_flow, max_flow);
        return AXT_EINVAL;
    }
    if (h_frame_ptr[0] != 0 || h_frame_ptr[n_frames] != n_det) {
        axt_set_error("axt_mcf_solve_dag: frame_ptr must run from 0 to n_det");
        return AXT_EINVAL;
    }
    Dag s;
    s.n = n_det; s.F = n_frames; s.fp = h_frame_ptr;
    s.obs = h_obs; s.entry = h_entry; s.exitc = h_exit; s.row_ptr = h_row_ptr; s.cost = h_cost; s.col = h_col;
    s.frame_of.resize(n_det);
    for (int t = 0; t < n_frames; ++t) {
        if (h_frame_ptr[t + 1] < h_frame_ptr[t]) { axt_set_error("axt_mcf_solve_dag: frame_ptr must not decrease"); return AXT_EINVAL; }
        for (int64_t k = h_frame_ptr[t]; k < h_frame_ptr[t + 1]; ++k) s.frame_of[k] = t;
    }
    int max_gap = 1;
    for (int k = 0; k < n_det; ++k)
        for (int64_t e = h_row_ptr[k]; e < h_row_ptr[k + 1]; ++e) {
            if (h_col[e] < 0 || h_col[e] >= n_det || s.frame_of[h_col[e]] <= s.frame_of[k]) {
                axt_set_error("axt_mcf_solve_dag: arc %lld of detection %d does not point forward in time", (long long)e, k);
                return AXT_EINVAL;
            }
            const int g = s.frame_of[h_col[e]] - s.frame_of[k];
            if (g > max_gap) max_gap = g;
        }
    s.succ.assign(n_det, NONE); s.pred_arc.assign(n_det, NONE); s.pred_tail.assign(n_det, -1);
    s.hU.resize(n_det); s.hV.resize(n_det); s.Bb.resize(n_det); s.bestV.resize(n_det);
    int flow = 0;
    int64_t total = 0;
    while (n_det > 0 && flow < max_flow) {
        if (!s.distances(max_gap)) { axt_set_error("axt_mcf_solve_dag: the sweeps did not converge"); return AXT_ERUNTIME; }
        int64_t best = INF;
        int k0 = -1;
        for (int k = 0; k < n_det; ++k) {
            if (s.pred_arc[k] == TERMINAL || s.hU[k] >= INF) continue;
            const int64_t v = h_entry[k] + s.hU[k];
            if (v < best) { best = v; k0 = k; }
        }
        if (k0 < 0) break;
        if (flow >= min_flow && best >= 0) break;
        if (!s.augment(k0)) { axt_set_error("axt_mcf_solve_dag: the distance labels do not describe a path"); return AXT_ERUNTIME; }
        total += best;
        ++flow;
    }
    if (getenv("AXT_MCF_DEBUG"))
        fprintf(stderr, "dag: n=%d frames=%d flow=%d sweeps=%zu steps=%zu step iterations=%zu\n", n_det, n_frames, flow,
                s.stat_sweeps, s.stat_steps, s.stat_iters);
    *n_tracks = 0;
    *total_cost = 0;
    for (int k = 0; k < n_det; ++k) { h_next[k] = -1; h_track[k] = -1; }
    if (flow < min_flow) return AXT_INFEASIBLE;
    int id = 0;
    for (int k = 0; k < n_det; ++k) {
        if (s.pred_arc[k] != TERMINAL) continue;
        int x = k;
        for (;;) {
            h_track[x] = id;
            const int32_t e = s.succ[x];
            if (e < 0) break;
            h_next[x] = h_col[e];
            x = h_col[e];
        }
        ++id;
    }
    *n_tracks = id;
    *total_cost = total;
    return AXT_OK;
}
