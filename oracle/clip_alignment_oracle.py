"""TEST INFRASTRUCTURE ONLY -- never imported by the product (vtc_amd/).

Scalar restatement of the reference's clip-alignment arithmetic and clip-grid rule, for checking
vtc_amd/pipeline/vtc_clip_alignment.py and vtc_amd/packaging/clips.py:

  merge / total / overlap of interval lists   /root/reference/src/pipeline/vtc_clip_alignment.py:84-119
  coverage per label and for all speech       :141-187
  IoU and best-IoU segment matching           :195-284  (5 s search margin, strict '>' keeps the first best)
  clip-relative -> absolute                   :58-76
  snap_to_grid                                /root/reference/src/packaging/clips.py:48-61

Segments are plain tuples (uid, onset, offset, label); clips are (clip_id, onset, offset, label) plus a
{clip_id: (uid, abs_onset)} map.  The reference's own tests do not cover this module (no fixtures exist for it), so the
restatement is pinned only by hand-worked cases in tests/test_clip_alignment.py: "parity unpinned" beyond those.
"""

from __future__ import annotations

LABELS = ["KCHI", "OCH", "MAL", "FEM"]
CHUNK_STEP_S = 63_680 / 16_000


def snap_to_grid(t, file_duration):
    return max(0.0, min(round(t / CHUNK_STEP_S) * CHUNK_STEP_S, file_duration))


def to_absolute(clip_segments, clip_map):
    out = []
    for clip_id, on, off, label in clip_segments:
        if clip_id in clip_map:                      # inner join
            uid, base = clip_map[clip_id]
            out.append((uid, on + base, off + base, label))
    return out


def merge(ivs):
    out = []
    for on, off in sorted(ivs):
        if out and on <= out[-1][1]:
            out[-1][1] = max(out[-1][1], off)
        else:
            out.append([on, off])
    return [(a, b) for a, b in out]


def total(ivs):
    return sum(b - a for a, b in ivs)


def overlap(a, b):
    tot, i, j = 0.0, 0, 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] <= b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def _group(segments):
    g = {}
    for uid, on, off, label in segments:
        g.setdefault((uid, label), []).append((on, off))
    return g


def coverage(full, clip_abs):
    fg, cg = _group(full), _group(clip_abs)
    uids = {u for u, _ in fg} | {u for u, _ in cg}
    res = {}
    for key in LABELS + ["all_speech"]:
        both = only_f = only_c = 0.0
        for uid in uids:
            labels = LABELS if key == "all_speech" else [key]
            f = merge([iv for l in labels for iv in fg.get((uid, l), [])])
            c = merge([iv for l in labels for iv in cg.get((uid, l), [])])
            ov = overlap(f, c)
            both += ov
            only_f += total(f) - ov
            only_c += total(c) - ov
        res[key] = {"both_h": both / 3600, "only_full_h": only_f / 3600, "only_clip_h": only_c / 3600}
    return res


def iou(a_on, a_off, b_on, b_off):
    inter = max(0.0, min(a_off, b_off) - max(a_on, b_on))
    union = max(a_off, b_off) - min(a_on, b_on)
    return inter / union if union > 0 else 0.0


def matches(full, clip_abs, collar_s=0.5):
    fg, cg = _group(full), _group(clip_abs)
    res = {}
    for label in LABELS:
        on_e, off_e, ious = [], [], []
        n_full = n_clip = n_well = 0
        for uid in {u for u, l in list(fg) + list(cg) if l == label}:
            fs, cs = sorted(fg.get((uid, label), [])), sorted(cg.get((uid, label), []))
            n_full += len(fs)
            n_clip += len(cs)
            start = 0
            for f_on, f_off in fs:
                while start < len(cs) and cs[start][1] < f_on - 5.0:
                    start += 1
                best, best_c = 0.0, None
                for c_on, c_off in cs[start:]:
                    if c_on > f_off + 5.0:
                        break
                    v = iou(f_on, f_off, c_on, c_off)
                    if v > best:
                        best, best_c = v, (c_on, c_off)
                if best_c is not None:
                    on_e.append(best_c[0] - f_on)
                    off_e.append(best_c[1] - f_off)
                    ious.append(best)
                    n_well += abs(on_e[-1]) <= collar_s and abs(off_e[-1]) <= collar_s
        res[label] = {"n_full": n_full, "n_clip": n_clip, "n_matched": len(ious), "n_well_matched": n_well,
                      "onset_errors": on_e, "offset_errors": off_e, "ious": ious}
    return res
