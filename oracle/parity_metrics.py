"""Parity metrics shared by tests/ and tools/parity_report.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as
oracle/vtc_oracle.py: nothing under vtc_amd/ imports this).

north_star tolerances (BASELINE.json): per-frame logits within 2e-2 abs, per-frame label agreement >= 99.9 %, segment
onset/offset within one 20 ms frame.  The last one is measured on segments: a segment counts as reproduced when the other
side holds a segment of the same label whose onset AND offset both lie within one frame of it; the fraction is reported in
both directions, for the raw table (vtc_raw) and for the collar-merged one (vtc_merged, the shipped output:
src/pipeline/vtc.py:383 merge_segments_df(seg_df, 0.3, 0.1)).  A flipped frame label inside a run splits the raw segment
(no partner within one frame); the 0.3 s collar re-joins such splits in the merged table, where in turn a one-frame
spurious segment can bridge two groups that lie 0.3-0.6 s apart.
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
            "raw": boundary_match(seg_ref, seg_got), "merged": boundary_match(mrg_ref, mrg_got)}


