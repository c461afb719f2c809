"""Parity metrics shared by tests/ and tools/parity_report.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as
oracle/vtc_oracle.py: nothing under vtc_amd/ imports this).

north_star tolerances (BASELINE.json): per-frame logits within 2e-2 abs, per-frame label agreement >= 99.9 %, segment
onset/offset within one 20 ms frame.  The last one is measured on segments: a segment counts as reproduced when the other
side holds a segment of the same label whose onset AND offset both lie within one frame of it; the fraction is reported in
both directions, for the raw table (vtc_raw) and for the collar-merged one (vtc_merged, the shipped output:
src/pipeline/vtc.py:383 merge_segments_df(seg_df, 0.3, 0.1)).  A flipped frame label inside a run splits the raw segment
(no partner within one frame); the 0.3 s collar re-joins such splits in the merged table, where in turn a one-frame
spurious segment can bridge two groups that lie 0.3-0.6 s apart.

The exact form of the criterion (attribute()): the HIP path's tables differ from the oracle's ONLY where the oracle's own
decision is within the logit tolerance of the threshold.  A frame-label is "flipped" when the two masks differ there; a flip
implies |oracle logit - threshold| <= |logit error|.  A raw segment (frames i..j of a label) is a function of that label's
mask on [i-1, j+1] alone, so a raw segment with no flip in that range is reproduced exactly.  A merged segment (first frame
i, last frame j) is a function of the mask on [i-17, j+17]: two runs join when onset_next <= offset_prev + 0.3 s, i.e.
(i_next - j_prev) * 0.02 - 0.025 <= 0.3, i.e. i_next - j_prev <= 16, so identical masks out to 17 frames on either side
fix both the group's members and its isolation.  compare() therefore reports, for every unmatched segment of either
side and either table, the flipped frame-labels in its range and the oracle's |logit| there; "unattributed" must be 0.
"""
from __future__ import annotations

import numpy as np
import torch

from . import vtc_oracle as O

ONE_FRAME_S = 0.02
THR = {l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in O.DEFAULT_LABELS}


def boundary_match(want: list, have: list, tol: float = ONE_FRAME_S + 1e-9) -> dict:
    """Segments as dicts (onset, offset, label).  A segment of `want` is matched when `have` holds a segment of the same
    label whose onset AND offset both lie within one 20 ms frame of it (north_star: "segment onset/offset within one
    20 ms frame").  Returned: matched fraction in both directions and the counts."""
    def frac(a, b):
        if not a:
            return 1.0, 0
        by = {}
        for s in b:
            by.setdefault(s["label"], []).append((s["onset"], s["offset"]))
        arr = {l: np.array(sorted(v)) for l, v in by.items()}
        ok = 0
        for s in a:
            v = arr.get(s["label"])
            if v is None:
                continue
            i = np.searchsorted(v[:, 0], s["onset"] - tol)
            while i < len(v) and v[i, 0] <= s["onset"] + tol:
                if abs(v[i, 1] - s["offset"]) <= tol:
                    ok += 1
                    break
                i += 1
        return ok / len(a), len(a) - ok
    f_w, miss_w = frac(want, have)
    f_h, miss_h = frac(have, want)
    return {"oracle_matched": f_w, "gpu_matched": f_h, "oracle_unmatched": miss_w, "gpu_unmatched": miss_h,
            "n_oracle": len(want), "n_gpu": len(have)}


COLLAR_FRAMES = 17      # see the module docstring: runs up to 16 frames apart join under the 0.3 s collar


def _unmatched(a: list, b: list, tol: float = ONE_FRAME_S + 1e-9) -> list:
    """Segments of a with no partner in b (same label, onset and offset within tol)."""
    by = {}
    for s in b:
        by.setdefault(s["label"], []).append((s["onset"], s["offset"]))
    arr = {l: np.array(sorted(v)) for l, v in by.items()}
    out = []
    for s in a:
        v = arr.get(s["label"])
        ok = False
        if v is not None:
            i = np.searchsorted(v[:, 0], s["onset"] - tol)
            while i < len(v) and v[i, 0] <= s["onset"] + tol:
                if abs(v[i, 1] - s["offset"]) <= tol:
                    ok = True
                    break
                i += 1
        if not ok:
            out.append(s)
    return out


def attribute(ref: torch.Tensor, m_ref, m_got, tables: dict, labels=O.DEFAULT_LABELS, threshold_logit: float = 0.0) -> dict:
    """tables: {"raw": (oracle segments, gpu segments), "merged": (...)}.  For every segment of either side without a
    partner within one frame: the flipped frame-labels of its label inside the range that determines it (+-1 frame for the
    raw table, +-COLLAR_FRAMES for the merged one).  Returns counts, the number left unattributed (no flip in range: would
    be a decode / merge difference, not a precision one) and the largest oracle |logit - threshold| over all flips."""
    flips = (np.asarray(m_ref) != np.asarray(m_got))
    dist = (ref.to(torch.float32) - threshold_logit).abs().numpy()
    n = flips.shape[0]
    out = {"flipped_frame_labels": int(flips.sum()),
           "max_oracle_abs_logit_at_flips": float(dist[flips].max()) if flips.any() else 0.0}
    for name, (want, have) in tables.items():
        margin = 1 if name == "raw" else COLLAR_FRAMES
        rec = {"unmatched": 0, "attributed": 0, "unattributed": 0, "max_oracle_abs_logit_at_cause": 0.0, "examples": []}
        for side, segs in (("oracle", _unmatched(want, have)), ("gpu", _unmatched(have, want))):
            for s in segs:
                j = labels.index(s["label"])
                i0 = int(round(s["onset"] / ONE_FRAME_S))
                i1 = int(round((s["offset"] - O.RF_SIZE / O.SAMPLE_RATE) / ONE_FRAME_S))
                a, b = max(0, i0 - margin), min(n, i1 + margin + 1)
                idx = np.flatnonzero(flips[a:b, j]) + a
                rec["unmatched"] += 1
                if len(idx):
                    rec["attributed"] += 1
                    worst = float(dist[idx, j].max())
                    rec["max_oracle_abs_logit_at_cause"] = max(rec["max_oracle_abs_logit_at_cause"], worst)
                    if len(rec["examples"]) < 8:
                        rec["examples"].append({"side": side, "label": s["label"], "onset": s["onset"], "offset": s["offset"],
                                                "flipped_frames": [int(k) for k in idx[:6]],
                                                "oracle_logits_there": [round(float(ref[k, j]), 5) for k in idx[:6]]})
                else:
                    rec["unattributed"] += 1
        out[name] = rec
    return out


def compare(ref: torch.Tensor, got: torch.Tensor) -> dict:
    """ref / got: (n_frames, 4) fp32 CPU logits of the oracle and of the HIP path."""
    d = (got - ref).abs()
    m_ref, m_got = O.apply_thresholds(ref, THR), O.apply_thresholds(got, THR)
    agree = float((m_ref == m_got).float().mean())
    seg_ref = O.intervals_to_segments(O.create_intervals(m_ref), "f")
    seg_got = O.intervals_to_segments(O.create_intervals(m_got), "f")
    mrg_ref = O.merge_segments(seg_ref, 0.3, 0.1)      # src/pipeline/vtc.py:383: merge_segments_df(seg_df, 0.3, 0.1)
    mrg_got = O.merge_segments(seg_got, 0.3, 0.1)
    return {"frames": int(ref.shape[0]), "logit_std": float(ref.std()),
            "max_abs_logit_err": float(d.max()), "mean_abs_logit_err": float(d.mean()),
            "frame_label_agreement": agree, "flipped_frame_labels": int((m_ref != m_got).sum()),
            "raw": boundary_match(seg_ref, seg_got), "merged": boundary_match(mrg_ref, mrg_got),
            "attribution": attribute(ref, m_ref, m_got, {"raw": (seg_ref, seg_got), "merged": (mrg_ref, mrg_got)})}


