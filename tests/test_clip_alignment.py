"""Clip-alignment consumer and clip-grid rule: vectorised host code against the scalar oracle
and hand-worked cases.  CPU only."""

import csv

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

from oracle import clip_alignment_oracle as AO
from vtc_amd.packaging.clips import CHUNK_STEP_F, CHUNK_STEP_S, chunk_step_samples, snap_to_grid
from vtc_amd.pipeline import vtc_clip_alignment as A


def seg_table(rows, id_col="uid"):
    return pa.table({id_col: [r[0] for r in rows], "onset": pa.array([r[1] for r in rows], pa.float64()),
                     "offset": pa.array([r[2] for r in rows], pa.float64()), "label": [r[3] for r in rows]})


def random_segments(rng, uids, n, t_max=600.0, grid=None):
    rows = []
    for _ in range(n):
        on = rng.uniform(0, t_max)
        dur = rng.choice([0.05, 0.4, 1.0, 3.0, 12.0]) * rng.uniform(0.5, 1.5)
        if grid:                                   # coarse grid -> many exact ties, touching and nested intervals
            on, dur = round(on / grid) * grid, max(grid, round(dur / grid) * grid)
        rows.append((str(rng.choice(uids)), float(on), float(on + dur), str(rng.choice(AO.LABELS))))
    return rows


def test_grid_rule():
    assert CHUNK_STEP_F == 63680 and CHUNK_STEP_S == 3.98 and chunk_step_samples(199) == 63680 and chunk_step_samples(499) == 159680
    for t, dur in ((0.0, 100.0), (1.98, 100.0), (1.99, 100.0), (2.0, 100.0), (5.97, 100.0), (9.95, 100.0), (99.9, 100.0),
                   (250.0, 100.0), (-3.0, 100.0), (601.7, 3600.0)):
        assert snap_to_grid(t, dur) == AO.snap_to_grid(t, dur)
    assert snap_to_grid(2.0, 100.0) == 3.98 and snap_to_grid(1.98, 100.0) == 0.0
    assert snap_to_grid(1.99, 100.0) == 0.0 and snap_to_grid(5.97, 100.0) == 2 * 3.98          # ties go to the even multiple
    assert snap_to_grid(99.9, 100.0) == 25 * 3.98 and snap_to_grid(250.0, 100.0) == 100.0 and snap_to_grid(-3.0, 100.0) == 0.0
    assert snap_to_grid(30.0, 100.0, step_s=9.98) == 3 * 9.98


def test_interval_algebra_hand_cases():
    on, off = A.merge_intervals(np.array([5.0, 0.0, 1.0, 2.0, 9.0]), np.array([6.0, 1.0, 1.5, 3.0, 9.5]))
    assert list(on) == [0.0, 2.0, 5.0, 9.0] and list(off) == [1.5, 3.0, 6.0, 9.5]            # [0,1] and [1,1.5] touch -> fuse
    on, off = A.merge_intervals(np.array([0.0, 1.0, 2.0]), np.array([10.0, 2.0, 3.0]))        # nested
    assert list(on) == [0.0] and list(off) == [10.0]
    assert A.interval_total(np.array([0.0, 2.0]), np.array([1.5, 3.0])) == 2.5
    a = (np.array([0.0, 4.0, 10.0]), np.array([2.0, 8.0, 11.0]))
    b = (np.array([1.0, 5.0, 7.5, 11.0]), np.array([4.5, 6.0, 10.5, 12.0]))
    assert A.interval_overlap(*a, *b) == pytest.approx(1.0 + 0.5 + 1.0 + 0.5 + 0.5)
    assert A.interval_overlap(*a, np.empty(0), np.empty(0)) == 0.0
    assert AO.overlap(list(zip(*a)), list(zip(*b))) == pytest.approx(3.5)


@pytest.mark.parametrize("grid", [None, 0.5])
def test_coverage_and_matching_against_oracle(grid):
    rng = np.random.default_rng(11 if grid else 12)
    uids = ["recA", "recB", "recC"]
    full = random_segments(rng, uids, 700, grid=grid)
    # the clip run: most full segments again with boundary jitter, some dropped, some extra
    clip = [(u, on + rng.normal(0, 0.2), off + rng.normal(0, 0.2), l) for u, on, off, l in full if rng.random() < 0.85]
    clip = [(u, min(a, b), max(a, b) + (0.0 if grid else 1e-3), l) for u, a, b, l in clip] if not grid else \
        [(u, on, off, l) for u, on, off, l in full if rng.random() < 0.85]
    clip += random_segments(rng, uids + ["recD"], 120, grid=grid)
    ft, ct = seg_table(full), seg_table(clip)
    cov, ref = A.compute_coverage(ft, ct), AO.coverage(full, clip)
    assert set(cov) == set(ref)
    for k in ref:
        for f in ("both_h", "only_full_h", "only_clip_h"):
            assert cov[k][f] == pytest.approx(ref[k][f], rel=1e-9, abs=1e-12), (k, f)
    for collar in (0.5, 5.0):
        m, mr = A.compute_segment_matches(ft, ct, collar_s=collar), AO.matches(full, clip, collar_s=collar)
        for l in AO.LABELS:
            for f in ("n_full", "n_clip", "n_matched", "n_well_matched"):
                assert m[l][f] == mr[l][f], (l, f)
            # same matched pairs: compare as multisets (group iteration order over uids is a set order in both)
            got = sorted(zip(m[l]["onset_errors"].tolist(), m[l]["offset_errors"].tolist(), m[l]["ious"].tolist()))
            want = sorted(zip(mr[l]["onset_errors"], mr[l]["offset_errors"], mr[l]["ious"]))
            assert got == want, l


def test_main_reads_parquet_and_writes_the_reference_csvs(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "manifests" / "ds.csv").write_text("path\n")
    out = tmp_path / "output" / "ds"
    for d in ("vtc_merged", "vtc_clips/vtc_merged", "stats"):
        (out / d).mkdir(parents=True)
    full = [("recA", 10.0, 12.0, "KCHI"), ("recA", 100.0, 103.0, "FEM"), ("recA", 500.0, 501.0, "FEM"),
            ("recB", 5.0, 6.0, "MAL")]
    # clip c1 of recA starts at 95.52 s (24 grid steps): the FEM segment re-detected 0.1 s late; c0 covers the KCHI one exactly
    clips = [("c0", 10.0, 12.0, "KCHI"), ("c1", 4.58, 7.58, "FEM"), ("c2", 1.0, 2.0, "OCH"), ("orphan", 0.0, 1.0, "FEM")]
    stats = pa.table({"clip_id": ["c0", "c1", "c2"], "uid": ["recA", "recA", "recB"], "abs_onset": [0.0, 95.52, 0.0]})
    pq.write_table(seg_table(full[:2]), out / "vtc_merged" / "shard_0.parquet")
    pq.write_table(seg_table(full[2:]), out / "vtc_merged" / "shard_1.parquet")
    pq.write_table(seg_table(clips, "clip_id"), out / "vtc_clips" / "vtc_merged" / "shard_0.parquet")
    pq.write_table(stats, out / "stats" / "clip_stats.parquet")
    A.main("ds")
    cov = {r["label"]: r for r in csv.DictReader(open(out / "vtc_clip_alignment_coverage.csv"))}
    assert list(cov["KCHI"]) == ["label", "both_h", "only_full_h", "only_clip_h", "agreement_pct"]
    assert float(cov["KCHI"]["agreement_pct"]) == 100.0 and float(cov["MAL"]["agreement_pct"]) == 0.0
    assert float(cov["FEM"]["agreement_pct"]) == round(100 * 2.9 / 4.1, 2)         # 2.9 s shared of 3 + 1 + 0.1 s
    mt = {r["label"]: r for r in csv.DictReader(open(out / "vtc_clip_alignment_matches.csv"))}
    assert list(mt["FEM"]) == ["label", "n_full", "n_clip", "n_matched", "match_pct"] + [f"within_{c}s_pct" for c in A.COLLARS] + \
        ["onset_median_s", "offset_median_s"]
    assert (mt["FEM"]["n_full"], mt["FEM"]["n_clip"], mt["FEM"]["n_matched"], mt["FEM"]["match_pct"]) == ("2", "1", "1", "50.0")
    assert float(mt["FEM"]["onset_median_s"]) == 0.1 and mt["OCH"]["onset_median_s"] == "" and mt["KCHI"]["within_0.5s_pct"] == "100.0"
    text = capsys.readouterr().out
    assert "VTC Clip Alignment — ds" in text and "Total speech per label" in text
    (out / "stats" / "clip_stats.parquet").unlink()
    with pytest.raises(SystemExit):
        A.main("ds")
