#!/usr/bin/env python3
"""Full-file VTC vs VTC re-run on packaged clips -- the consumer that compares the two callers of the hot path
(mirrors `python -m src.pipeline.vtc_clip_alignment`, /root/reference/src/pipeline/vtc_clip_alignment.py).

Reads
    output/{dataset}/vtc_merged/*.parquet             full-file segments (uid, onset, offset, label)
    output/{dataset}/vtc_clips/vtc_merged/*.parquet   clip segments, clip-relative (clip_id, onset, offset, label)
    output/{dataset}/stats/clip_stats.parquet         clip_id, uid, abs_onset
and writes `vtc_clip_alignment_coverage.csv` and `vtc_clip_alignment_matches.csv` into output/{dataset}/ with the
reference's columns (:407-462).  Same two analyses: per-label coverage in hours (both / only full / only clip,
interval arithmetic, :141-187) and segment matching by best temporal IoU with onset/offset errors (:201-284).
Figures (src.plotting) are outside SURVEY.md section 8 and are not drawn.

The arithmetic runs on sorted numpy arrays per (uid, label) group instead of Python lists of tuples; results equal
the reference's formulas (oracle/clip_alignment_oracle.py) -- counts and matched pairs exactly, hour totals to
summation-order rounding.

Usage:
    python -m vtc_amd.pipeline.vtc_clip_alignment seedlings_10
"""

from __future__ import annotations

import csv
import logging
import sys
from pathlib import Path

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

from ..utils import get_dataset_paths

logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s", datefmt="%H:%M:%S", stream=sys.stdout)
logger = logging.getLogger("vtc_align")

LABELS = ["KCHI", "OCH", "MAL", "FEM"]
COLLARS = [0.5, 1.0, 2.0, 3.0, 5.0]


def _load_parquet_dir(path: Path) -> pa.Table:
    files = sorted(Path(path).glob("*.parquet"))
    if not files:
        raise FileNotFoundError(f"No parquet files in {path}")
    return pa.concat_tables([pq.read_table(f) for f in files], promote_options="default")


def _to_absolute(clip_tbl: pa.Table, clip_stats: pa.Table) -> pa.Table:
    """Clip-relative -> absolute time: inner join on clip_id, shift by the clip's abs_onset (:58-76)."""
    joined = clip_tbl.select(["clip_id", "onset", "offset", "label"]).join(
        clip_stats.select(["clip_id", "uid", "abs_onset"]), keys="clip_id", join_type="inner")
    base = joined.column("abs_onset").to_numpy(zero_copy_only=False).astype(np.float64)
    return pa.table({"uid": joined.column("uid"),
                     "onset": joined.column("onset").to_numpy(zero_copy_only=False) + base,
                     "offset": joined.column("offset").to_numpy(zero_copy_only=False) + base,
                     "label": joined.column("label")})


# ------------------------------------------------------------------------------------------------ interval algebra
def merge_intervals(on: np.ndarray, off: np.ndarray):
    """Union of closed intervals as sorted, disjoint (on, off) arrays; touching intervals fuse (:84-97)."""
    if len(on) == 0:
        return np.empty(0), np.empty(0)
    order = np.lexsort((off, on))
    on, off = on[order], off[order]
    reach = np.maximum.accumulate(off)
    starts = np.ones(len(on), dtype=bool)
    starts[1:] = on[1:] > reach[:-1]
    first = np.flatnonzero(starts)
    last = np.append(first[1:], len(on)) - 1
    return on[first], reach[last]


def interval_total(on: np.ndarray, off: np.ndarray) -> float:
    return float((off - on).sum())


def interval_overlap(a_on, a_off, b_on, b_off) -> float:
    """Total intersection of two sorted disjoint interval sets (:104-119).  Interval i of `a` can only meet the b's
    from the first one ending after a_on[i] to the last one starting before a_off[i]; clip and sum."""
    if len(a_on) == 0 or len(b_on) == 0:
        return 0.0
    lo = np.searchsorted(b_off, a_on, side="right")
    hi = np.searchsorted(b_on, a_off, side="left")
    total = 0.0
    for i in np.flatnonzero(hi > lo):
        seg = np.minimum(a_off[i], b_off[lo[i]:hi[i]]) - np.maximum(a_on[i], b_on[lo[i]:hi[i]])
        total += float(seg[seg > 0].sum())
    return total


def _groups(tbl: pa.Table) -> dict:
    """{(uid, label): (onset array, offset array)} in input order."""
    uid = tbl.column("uid").to_pylist()
    lab = tbl.column("label").to_pylist()
    on = tbl.column("onset").to_numpy(zero_copy_only=False).astype(np.float64)
    off = tbl.column("offset").to_numpy(zero_copy_only=False).astype(np.float64)
    index: dict = {}
    for i, key in enumerate(zip(uid, lab)):
        index.setdefault(key, []).append(i)
    return {k: (on[v], off[v]) for k, v in index.items()}


_EMPTY = (np.empty(0), np.empty(0))


def compute_coverage(full_tbl: pa.Table, clip_abs_tbl: pa.Table) -> dict:
    """Per label and for all speech (union over labels): hours detected by both runs, only by the full-file run, only
    by the clip run (:141-187)."""
    fg, cg = _groups(full_tbl), _groups(clip_abs_tbl)
    uids = {u for u, _ in fg} | {u for u, _ in cg}
    acc = {k: np.zeros(3) for k in LABELS + ["all_speech"]}

    def add(key, f, c):
        fm, cm = merge_intervals(*f), merge_intervals(*c)
        ov = interval_overlap(*fm, *cm)
        acc[key] += (ov, interval_total(*fm) - ov, interval_total(*cm) - ov)

    for uid in uids:
        for label in LABELS:
            add(label, fg.get((uid, label), _EMPTY), cg.get((uid, label), _EMPTY))
        f_all = [fg[(uid, l)] for l in LABELS if (uid, l) in fg]
        c_all = [cg[(uid, l)] for l in LABELS if (uid, l) in cg]
        cat = lambda parts: (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])) if parts else _EMPTY
        add("all_speech", cat(f_all), cat(c_all))
    return {k: {"both_h": v[0] / 3600, "only_full_h": v[1] / 3600, "only_clip_h": v[2] / 3600} for k, v in acc.items()}


def compute_segment_matches(full_tbl: pa.Table, clip_abs_tbl: pa.Table, collar_s: float = 0.5) -> dict:
    """For every full-file segment the clip segment of the same (uid, label) with the highest temporal IoU (first one
    in (onset, offset) order on ties); matched iff IoU > 0, well matched iff both boundaries agree within collar_s
    (:201-284).  IoU > 0 needs a true overlap, so only overlapping candidates are scored."""
    fg, cg = _groups(full_tbl), _groups(clip_abs_tbl)
    out = {}
    for label in LABELS:
        n_full = n_clip = n_well = 0
        on_err, off_err, ious = [], [], []
        for uid in {u for u, l in fg if l == label} | {u for u, l in cg if l == label}:
            f_on, f_off = fg.get((uid, label), _EMPTY)
            c_on, c_off = cg.get((uid, label), _EMPTY)
            n_full += len(f_on)
            n_clip += len(c_on)
            if len(f_on) == 0 or len(c_on) == 0:
                continue
            fo = np.lexsort((f_off, f_on))
            co = np.lexsort((c_off, c_on))
            f_on, f_off, c_on, c_off = f_on[fo], f_off[fo], c_on[co], c_off[co]
            reach = np.maximum.accumulate(c_off)
            lo = np.searchsorted(reach, f_on, side="right")      # everything before ends at or before f_on
            hi = np.searchsorted(c_on, f_off, side="left")       # everything from here starts at or after f_off
            for i in np.flatnonzero(hi > lo):
                a, b = c_on[lo[i]:hi[i]], c_off[lo[i]:hi[i]]
                overlap = np.maximum(0.0, np.minimum(f_off[i], b) - np.maximum(f_on[i], a))
                union = np.maximum(f_off[i], b) - np.minimum(f_on[i], a)
                iou = np.where(union > 0, overlap / np.where(union > 0, union, 1.0), 0.0)
                j = int(np.argmax(iou))
                if iou[j] > 0:
                    oe, fe = float(a[j] - f_on[i]), float(b[j] - f_off[i])
                    on_err.append(oe)
                    off_err.append(fe)
                    ious.append(float(iou[j]))
                    n_well += abs(oe) <= collar_s and abs(fe) <= collar_s
        out[label] = {"n_full": n_full, "n_clip": n_clip, "n_matched": len(ious), "n_well_matched": int(n_well),
                      "onset_errors": np.array(on_err), "offset_errors": np.array(off_err), "ious": np.array(ious)}
    return out


# ------------------------------------------------------------------------------------------------------- reporting
def coverage_rows(coverage: dict) -> list[dict]:
    rows = []
    for lbl in LABELS + ["all_speech"]:
        c = coverage[lbl]
        total = c["both_h"] + c["only_full_h"] + c["only_clip_h"]
        rows.append({"label": lbl, "both_h": round(c["both_h"], 2), "only_full_h": round(c["only_full_h"], 2),
                     "only_clip_h": round(c["only_clip_h"], 2),
                     "agreement_pct": round(100 * c["both_h"] / total, 2) if total > 0 else 0})
    return rows


def _within(m: dict, collar: float) -> int:
    return int(np.sum((np.abs(m["onset_errors"]) <= collar) & (np.abs(m["offset_errors"]) <= collar)))


def match_rows(matches: dict) -> list[dict]:
    rows = []
    for lbl in LABELS:
        m = matches[lbl]
        n = m["n_full"]
        row = {"label": lbl, "n_full": n, "n_clip": m["n_clip"], "n_matched": m["n_matched"],
               "match_pct": round(100 * m["n_matched"] / n, 2) if n > 0 else 0}
        for c in COLLARS:
            row[f"within_{c}s_pct"] = round(100 * _within(m, c) / n, 2) if n > 0 else 0
        has = len(m["onset_errors"]) > 0
        row["onset_median_s"] = round(float(np.median(m["onset_errors"])), 4) if has else None
        row["offset_median_s"] = round(float(np.median(m["offset_errors"])), 4) if has else None
        rows.append(row)
    return rows


def _write_csv(path: Path, rows: list[dict]) -> None:
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        for r in rows:
            w.writerow({k: ("" if v is None else v) for k, v in r.items()})


def label_totals(full_tbl: pa.Table, clip_tbl: pa.Table) -> dict:
    def hours(tbl, lbl):
        lab = np.array(tbl.column("label").to_pylist(), dtype=object)
        d = tbl.column("offset").to_numpy(zero_copy_only=False) - tbl.column("onset").to_numpy(zero_copy_only=False)
        return float(d[lab == lbl].sum()) / 3600

    totals = {lbl: {"full_h": hours(full_tbl, lbl), "clip_h": hours(clip_tbl, lbl)} for lbl in LABELS}
    totals["all_speech"] = {k: sum(t[k] for t in totals.values()) for k in ("full_h", "clip_h")}
    return totals


def main(dataset: str) -> None:
    paths = get_dataset_paths(dataset)
    orig_dir = paths.output / "vtc_merged"
    clip_dir = paths.output / "vtc_clips" / "vtc_merged"
    clip_stats_path = paths.output / "stats" / "clip_stats.parquet"
    for p in (orig_dir, clip_dir, clip_stats_path):
        if not p.exists():
            logger.error(f"Missing: {p}")
            sys.exit(1)
    logger.info(f"Dataset : {dataset}")
    full_tbl = _load_parquet_dir(orig_dir)
    clip_tbl = _load_parquet_dir(clip_dir)
    clip_stats = pq.read_table(clip_stats_path)
    logger.info(f"  Full-file segments : {full_tbl.num_rows:,}")
    logger.info(f"  Clip segments      : {clip_tbl.num_rows:,}")
    logger.info(f"  Clips              : {clip_stats.num_rows:,}")
    clip_abs = _to_absolute(clip_tbl, clip_stats)
    logger.info(f"  Clip segments (absolute) : {clip_abs.num_rows:,}")
    coverage = compute_coverage(full_tbl, clip_abs)
    matches = compute_segment_matches(full_tbl, clip_abs, collar_s=max(COLLARS))

    print()
    print("=" * 72)
    print(f"VTC Clip Alignment — {dataset}")
    print("=" * 72)
    print("\n── Coverage (hours) ──")
    print(f"  {'Label':<14} {'Both':>8} {'Only full':>10} {'Only clip':>10} {'Agree%':>8}")
    print("  " + "-" * 54)
    for lbl in LABELS + ["all_speech"]:
        c = coverage[lbl]
        total = c["both_h"] + c["only_full_h"] + c["only_clip_h"]
        agree = 100 * c["both_h"] / total if total > 0 else 0
        print(f"  {lbl:<14} {c['both_h']:>8.1f} {c['only_full_h']:>10.1f} {c['only_clip_h']:>10.1f} {agree:>7.1f}%")
    print("\n── Segment matching (multi-collar) ──")
    print(f"  {'Label':<8} {'N_full':>8} {'Matched':>8}" + "".join(f" {'±' + str(c) + 's':>8}" for c in COLLARS))
    print("  " + "-" * (32 + 9 * len(COLLARS)))
    for lbl in LABELS:
        m = matches[lbl]
        n = m["n_full"]
        print(f"  {lbl:<8} {n:>8,} {m['n_matched']:>8,}" +
              "".join(f" {(100 * _within(m, c) / n if n > 0 else 0):>7.1f}%" for c in COLLARS))
    print("\n── Boundary errors (matched segments, seconds) ──")
    print(f"  {'Label':<10} {'Onset med':>10} {'Onset IQR':>10} {'Offset med':>11} {'Offset IQR':>11}")
    print("  " + "-" * 56)
    for lbl in LABELS:
        m = matches[lbl]
        if len(m["onset_errors"]) > 0:
            iqr = lambda v: np.percentile(v, 75) - np.percentile(v, 25)
            print(f"  {lbl:<10} {np.median(m['onset_errors']):>+10.3f} {iqr(m['onset_errors']):>10.3f}"
                  f" {np.median(m['offset_errors']):>+11.3f} {iqr(m['offset_errors']):>11.3f}")
        else:
            print(f"  {lbl:<10} {'n/a':>10} {'n/a':>10} {'n/a':>11} {'n/a':>11}")

    out_dir = paths.output
    cov_csv = out_dir / "vtc_clip_alignment_coverage.csv"
    _write_csv(cov_csv, coverage_rows(coverage))
    logger.info(f"Saved -> {cov_csv}")
    match_csv = out_dir / "vtc_clip_alignment_matches.csv"
    _write_csv(match_csv, match_rows(matches))
    logger.info(f"Saved -> {match_csv}")

    totals = label_totals(full_tbl, clip_tbl)
    print("\n── Total speech per label (hours) ──")
    print(f"  {'Label':<14} {'Full-file':>10} {'Clip':>10} {'Diff%':>8}")
    print("  " + "-" * 44)
    for lbl in LABELS + ["all_speech"]:
        t = totals[lbl]
        diff = 100 * (t["clip_h"] - t["full_h"]) / t["full_h"] if t["full_h"] > 0 else 0
        print(f"  {lbl:<14} {t['full_h']:>10.2f} {t['clip_h']:>10.2f} {diff:>+7.2f}%")
    logger.info("figures are not drawn by this build (plotting is outside the VTC path)")


if __name__ == "__main__":
    if len(sys.argv) < 2:
        print("Usage: python -m vtc_amd.pipeline.vtc_clip_alignment <dataset>", file=sys.stderr)
        sys.exit(1)
    main(sys.argv[1])
