"""CPU: host-side mirror of the reference interface (no GPU compute), the C-ABI library's exports,
and the N>1 sharding path over gloo (world_size 2)."""

import ctypes
import json
import math
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest
import torch

from oracle import vtc_oracle as O
from vtc_amd import lib, synth, utils
from vtc_amd.core import intervals as I
from vtc_amd.core import metadata as M
from vtc_amd.segma_compat import install_as_segma
from vtc_amd.segma_compat.config import load_config
from vtc_amd.segma_compat.inference import logit_bounds
from vtc_amd.segma_compat.models import ConvSettings, Models, normalise_state_dict
from vtc_amd.segma_compat.utils.conversions import frames_to_seconds
from vtc_amd.segma_compat.utils.encoders import MultiLabelEncoder

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.loads((golden_dir / "golden_kat.json").read_text())


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "vtc_amd.h").read_text()
    declared = set(re.findall(r"\b(vtc_[a-z_]+)\s*\(", header))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    so = lib.load()
    for name in declared:
        assert getattr(so, name) is not None


def test_no_gpu_calls_fail_loudly_not_silently():
    so = lib.load()
    assert so.vtc_frames_for_samples(399) == 0 and so.vtc_frames_for_samples(400) == 1
    assert so.vtc_frames_for_samples(64000) == 199 and so.vtc_frames_for_samples(160000) == 499
    assert so.vtc_profile_class_name(2) == b"gemm_mfma"
    if not torch.cuda.is_available():
        spec = lib.VtcSpec(abi_version=lib.ABI_VERSION, n_labels=4, window_samples=64000, max_batch=1, mfma_dtype=1)
        h = ctypes.c_void_p()
        rc = so.vtc_create(ctypes.byref(spec), ctypes.byref(lib.VtcWeights()), 0, ctypes.byref(h))
        assert rc == lib.ERR_NO_GPU and not h.value
        assert b"HIP device" in so.vtc_last_error(None)
        from vtc_amd.engine import VtcEngine
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            VtcEngine(synth.make_weights(0), 4, 4.0)


def test_bad_spec_rejected():
    so = lib.load()
    h = ctypes.c_void_p()
    spec = lib.VtcSpec(abi_version=99, n_labels=4, window_samples=64000, max_batch=1, mfma_dtype=1)
    assert so.vtc_create(ctypes.byref(spec), ctypes.byref(lib.VtcWeights()), 0, ctypes.byref(h)) == lib.ERR_ARG
    spec = lib.VtcSpec(abi_version=lib.ABI_VERSION, n_labels=4, window_samples=700000, max_batch=1, mfma_dtype=1)
    assert so.vtc_create(ctypes.byref(spec), ctypes.byref(lib.VtcWeights()), 0, ctypes.byref(h)) == lib.ERR_ARG
    assert b"frames" in so.vtc_last_error(None)


# ------------------------------------------------------------------ weights / synth
def test_synth_weights_equal_oracle_weights():
    a, b = synth.make_weights(seed=5), O.make_weights(seed=5)
    assert set(a) == set(b)
    for k in a:
        np.testing.assert_allclose(a[k], b[k].numpy(), rtol=2e-6, atol=1e-7, err_msg=k)


def test_synth_audio_is_seeded_and_bounded():
    x, y = synth.synth_audio(2.0, seed=3), synth.synth_audio(2.0, seed=3)
    assert x.dtype == np.float32 and len(x) == 32000 and np.array_equal(x, y)
    assert np.abs(x).max() < 1.0 and not np.array_equal(x, synth.synth_audio(2.0, seed=4))


# ------------------------------------------------------------------ segma surface
def test_conv_settings_and_encoder():
    cs = ConvSettings()
    assert cs.rf_step == 320 and cs.rf_size == 400
    assert [cs.n_windows(n) for n in (399, 400, 4800, 64000, 160000)] == [0, 1, 14, 199, 499]
    enc = MultiLabelEncoder(labels=["FEM", "MAL", "KCHI", "OCH"])
    assert enc.n_labels == 4 and enc.inv_transform(2) == "KCHI" and enc.transform("OCH") == 3
    with pytest.raises(ValueError):
        MultiLabelEncoder(labels=["A", "A"])
    assert frames_to_seconds(16000) == 1.0


def test_config_and_checkpoint_loading(tmp_path):
    cfg_path = tmp_path / "config.yml"
    cfg_path.write_text("model:\n  name: surgical_hydra\n  extra: 1\ndata:\n  classes: [KCHI, OCH, MAL, FEM]\n"
                        "audio:\n  chunk_duration_s: 10\n  sample_rate: 16000\ntrain:\n  lr: 1e-3\n")
    cfg = load_config(cfg_path)
    assert cfg.data.classes == ["KCHI", "OCH", "MAL", "FEM"] and cfg.audio.chunk_duration_s == 10.0
    assert cfg.model.name == "surgical_hydra" and cfg.engine.mfma_dtype == "f16"
    enc = MultiLabelEncoder(labels=cfg.data.classes)
    m = Models[cfg.model.name].load_from_checkpoint(checkpoint_path="synthetic:3", label_encoder=enc, config=cfg, train=False)
    assert m.eval() is m and m.to("cpu") is m and m.conv_settings.rf_step == 320
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.engine(10.0, 4)
    # wrapped Lightning-style checkpoint, safe loader
    sd = {"model.wav_encoder." + k: torch.from_numpy(v) for k, v in synth.make_weights(1).items() if not k.startswith("head.")}
    sd["model.classifier.weight"] = torch.zeros(4, 768)
    sd["model.classifier.bias"] = torch.ones(4)
    torch.save({"state_dict": sd}, tmp_path / "best.ckpt")
    m2 = Models["whatever"].load_from_checkpoint(checkpoint_path=tmp_path / "best.ckpt", label_encoder=enc, config=cfg)
    assert "feature_extractor.conv_layers.0.conv.weight" in m2.state_dict_host
    assert tuple(m2.state_dict_host["head.weight"].shape) == (4, 768)
    with pytest.raises(KeyError):
        normalise_state_dict({"foo": np.zeros(3)}, 4)
    # the same weights under torchaudio's and fairseq's HuBERT names come out under the HF names the engine reads
    import re
    hf = {k: v for k, v in synth.make_weights(2).items()}
    ta = {"wrapped.model." + (re.sub(r"^encoder\.", "encoder.transformer.", k) if k.startswith("encoder.") else
                              re.sub(r"^feature_projection\.", "encoder.feature_projection.", k)): v
          for k, v in hf.items() if not k.startswith("head.")}
    ta["wrapped.heads.weight"], ta["wrapped.heads.bias"] = hf["head.weight"], hf["head.bias"]
    fs = {}
    for k, v in hf.items():
        k = re.sub(r"^feature_extractor\.conv_layers\.(\d+)\.conv\.", r"feature_extractor.conv_layers.\1.0.", k)
        k = re.sub(r"^feature_extractor\.conv_layers\.0\.layer_norm\.", "feature_extractor.conv_layers.0.2.", k)
        k = re.sub(r"^feature_projection\.layer_norm\.", "layer_norm.", k)
        k = re.sub(r"^feature_projection\.projection\.", "post_extract_proj.", k)
        k = re.sub(r"^encoder\.pos_conv_embed\.conv\.", "encoder.pos_conv.0.", k)
        k = re.sub(r"\.attention\.", ".self_attn.", k)
        k = re.sub(r"^(encoder\.layers\.\d+)\.layer_norm\.", r"\1.self_attn_layer_norm.", k)
        k = re.sub(r"\.feed_forward\.intermediate_dense\.", ".fc1.", k)
        k = re.sub(r"\.feed_forward\.output_dense\.", ".fc2.", k)
        fs[("w2v." + k) if not k.startswith("head.") else ("cls." + k[5:])] = v
    for other in (ta, fs):
        got = normalise_state_dict(other, 4)
        assert set(hf) <= set(got), sorted(set(hf) - set(got))[:5]
        assert all(np.array_equal(got[k], hf[k]) for k in hf)


def test_install_as_segma_matches_reference_imports():
    install_as_segma()
    from segma.config import load_config as lc            # noqa: F401  (reference src/pipeline/vtc.py:41-47)
    from segma.config.base import Config                  # noqa: F401
    from segma.inference import apply_model_on_audio, apply_thresholds, create_intervals  # noqa: F401
    from segma.models import Models as Ms                 # noqa: F401
    from segma.utils.encoders import MultiLabelEncoder as E  # noqa: F401
    from segma.utils.conversions import frames_to_seconds as f2s  # reference src/core/intervals.py:10
    assert f2s(320) == 0.02


def test_logit_bounds_reproduce_prob_space_thresholds():
    """x >= lo  <=>  sigmoid_f32(x) >= p.  torch's CPU sigmoid differs by an ulp between its vector and tail
    paths, so exact agreement is asserted away from the boundary and the boundary is pinned to <= 4 ulp."""
    rng = np.random.default_rng(0)
    for p in (0.5, 0.3, 0.71, 0.05, 0.999):
        lo, hi = logit_bounds({"A": {"lower_bound": p, "upper_bound": 1.0}}, ["A"])
        assert hi[0] == np.inf
        exact = math.log(np.float32(p) / (1 - np.float64(np.float32(p))))
        tol = 4 * np.spacing(np.float32(abs(exact))) + 4e-7 + 4 * np.spacing(np.float32(p)) / (p * (1 - p))
        assert abs(lo[0] - exact) <= tol
        xs = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 3,
                             (exact + rng.uniform(-1e-3, 1e-3, 20000)).astype(np.float32)])
        xs = xs[np.abs(xs - lo[0]) > tol]
        want = (torch.sigmoid(torch.from_numpy(xs)) >= torch.tensor(p, dtype=torch.float32)).numpy()
        assert np.array_equal(xs >= lo[0], want)
    lo, hi = logit_bounds({"A": {"lower_bound": 0.5, "upper_bound": 1.0}}, ["A"])
    assert lo[0] < 0 and lo[0] > -1e-6          # sigmoid(0) == 0.5 exactly: logit 0 is active
    lo, hi = logit_bounds({"A": {"lower_bound": 0.2, "upper_bound": 0.8}}, ["A"])
    xs = np.round((rng.standard_normal(5000) * 2) * 1024) / 1024
    xs = xs.astype(np.float32)
    pr = torch.sigmoid(torch.from_numpy(xs))
    want = ((pr >= torch.tensor(0.2)) & (pr <= torch.tensor(0.8))).numpy()
    assert np.array_equal((xs >= lo[0]) & (xs <= hi[0]), want)


# ------------------------------------------------------------------ intervals / metadata / utils
def test_intervals_match_oracle_and_kat(kat):
    for case in kat["segments"]:
        iv = [tuple(i) for i in case["intervals"]]
        assert I.intervals_to_segments(iv, case["uid"]) == case["rows"] == O.intervals_to_segments(iv, case["uid"])
    # reference tests/test_intervals.py cases
    assert I.merge_pairs([]) == []
    assert I.merge_pairs([(0.0, 2.0), (1.0, 3.0), (5.0, 6.0)]) == [(0.0, 3.0), (5.0, 6.0)]
    assert I.merge_pairs([(0.0, 1.0), (1.0, 2.0)]) == [(0.0, 2.0)]
    assert I.merge_pairs([(0.0, 10.0), (2.0, 5.0)]) == [(0.0, 10.0)]
    assert I.merge_pairs([(3.0, 4.0), (0.0, 1.0), (2.0, 3.5)]) == [(0.0, 1.0), (2.0, 4.0)]
    assert I.total_duration([(0.0, 1.0), (3.0, 5.0)]) == 3.0
    assert I.intervals_to_pairs([(0, 16000, "KCHI"), (8000, 24000, "FEM")]) == [(0.0, 1.5)]
    assert I.intervals_to_segments([], "t") == []
    rng = np.random.default_rng(2)
    iv = [(int(a) * 320, int(a + b) * 320 + 400, "FEM") for a, b in zip(rng.integers(0, 10**6, 500), rng.integers(0, 300, 500))]
    assert I.intervals_to_segments(iv, "u") == O.intervals_to_segments(iv, "u")


def test_metadata_rows_match_reference_goldens(golden_dir):
    g = json.loads((golden_dir / "golden_meta.json").read_text())
    for case in g["meta_rows"]:
        assert M.vtc_meta_row(*case["args"]) == case["row"]
        assert list(M.vtc_meta_row(*case["args"])) == list(case["row"])   # same key order -> same column order
    for case in g["error_rows"]:
        row = M.vtc_error_row(*case["args"])
        assert set(row) == set(case["row"])
        for k, v in case["row"].items():
            assert (math.isnan(row[k]) if v is None else row[k] == v)


def test_shard_list(kat):
    s = kat["shard"]
    items = list(range(s["n"]))
    assert [utils.shard_list(items, i, s["count"]) for i in range(s["count"])] == s["slices"]
    for n in (0, 1, 7, 8, 9, 100):
        for c in (1, 3, 8):
            parts = [utils.shard_list(list(range(n)), i, c) for i in range(c)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
            assert parts == [O.shard_list(list(range(n)), i, c) for i in range(c)]
    with pytest.raises(ValueError):
        utils.shard_list([1], 3, 3)


def test_balanced_shards_cover_and_balance():
    rng = np.random.default_rng(0)
    costs = np.exp(rng.uniform(0, np.log(57600), 500)).tolist()
    items = [f"f{i}" for i in range(500)]
    parts = utils.balanced_shards(items, costs, 8)
    assert sorted(sum(parts, [])) == sorted(items)
    loads = [sum(costs[int(x[1:])] for x in p) for p in parts]
    assert max(loads) / (sum(loads) / 8) < 1.15
    naive = [sum(costs[int(x[1:])] for x in utils.shard_list(items, i, 8)) for i in range(8)]
    assert max(loads) <= max(naive)


def _rows(kat_rows):
    return [dict(uid=u, label=l, onset=a, offset=b, duration=round(b - a, 3)) for u, l, a, b in kat_rows]


def test_merge_segments_kat_and_oracle(kat):
    m = kat["merge"]
    got = utils.merge_segments(utils.segments_table(_rows(m["rows"])), m["collar"], m["min_on"])
    assert got.column_names == ["uid", "onset", "offset", "duration", "label"]
    rows = got.to_pylist()
    assert [[r["uid"], r["label"], r["onset"], r["offset"], r["duration"]] for r in rows] == m["expected"]
    # randomised agreement with the oracle's loop restatement
    rng = np.random.default_rng(1)
    rnd = []
    for uid in ("u1", "u2", "u3"):
        for lab in ("FEM", "MAL", "KCHI", "OCH"):
            t = 0.0
            for _ in range(int(rng.integers(0, 60))):
                t += float(rng.choice([0.02, 0.1, 0.29, 0.3, 0.31, 0.5, 2.0]))
                d = float(rng.choice([0.02, 0.045, 0.085, 0.105, 0.5, 3.985]))
                rnd.append(dict(uid=uid, label=lab, onset=round(t, 3), offset=round(t + d, 3), duration=round(d, 3)))
                t += d * float(rng.choice([0.5, 1.0]))
    rng.shuffle(rnd)
    got = utils.merge_segments(utils.segments_table(rnd), 0.3, 0.1).to_pylist()
    assert got == O.merge_segments(rnd, 0.3, 0.1)
    empty = utils.segments_table([])
    assert utils.merge_segments(empty).num_rows == 0


def test_atomic_write_resume_and_manifest(tmp_path, monkeypatch):
    rows = [M.vtc_meta_row("a", 0.5, [], 0.1, 0.05), M.vtc_error_row("b", "bad file")]
    utils.atomic_write_parquet(utils.meta_table(rows), tmp_path / "vtc_meta" / "shard_0.parquet")
    utils.atomic_write_parquet(utils.meta_table([M.vtc_meta_row("c", 0.5, [], 0.1, 0.05)]), tmp_path / "vtc_meta" / "shard_1.parquet")
    assert not list((tmp_path / "vtc_meta").glob("*.tmp"))
    assert utils.load_completed_ids(tmp_path / "vtc_meta", "uid", "shard_*.parquet") == {"a", "b", "c"}   # error rows count
    assert utils.load_completed_ids(tmp_path / "nope") == set()
    t = pq.read_table(tmp_path / "vtc_meta" / "shard_0.parquet")
    assert t.schema.names == list(utils.META_SCHEMA.names)
    assert pq.ParquetFile(tmp_path / "vtc_meta" / "shard_0.parquet").metadata.row_group(0).column(0).compression == "ZSTD"
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "manifests" / "ds.csv").write_text("path,extra\n/x/a.wav,1\n/x/b.wav,2\n/x/c.wav,3\n")
    paths = utils.get_dataset_paths("ds")
    assert paths.manifest == Path("manifests/ds.csv") and paths.output == Path("output/ds")
    man = utils.load_manifest(paths.manifest)
    assert utils.manifest_paths(man) == ["/x/a.wav", "/x/b.wav", "/x/c.wav"]
    assert utils.sample_manifest(man, None).num_rows == 3 and utils.sample_manifest(man, 2).num_rows == 2
    assert utils.sample_manifest(man, 2).equals(utils.sample_manifest(man, 2))
    with pytest.raises(FileNotFoundError):
        utils.get_dataset_paths("missing")
    rec = utils.log_benchmark("vtc", "ds", 3, 1.5, total_bytes=3000, extra={"shard_id": 0})
    assert rec["bytes_per_second"] == 2000.0 and json.loads((tmp_path / "logs/benchmarks.jsonl").read_text())["step"] == "vtc"


def test_audio_reader(tmp_path):
    from vtc_amd.audio import read_audio, write_wav
    from scipy.io import wavfile
    x = synth.synth_audio(0.5, seed=1)
    write_wav(tmp_path / "a.wav", x)
    y = read_audio(tmp_path / "a.wav")
    assert y.dtype == np.float32 and len(y) == 8000 and np.abs(y - x).max() <= 1.0 / 32768 + 1e-7
    st = np.stack([x, -x], axis=1)
    wavfile.write(tmp_path / "st.wav", 16000, (st * 32767).astype(np.int16))
    assert np.abs(read_audio(tmp_path / "st.wav")).max() <= 1e-4          # channels averaged
    wavfile.write(tmp_path / "r8.wav", 8000, (x[::2] * 32767).astype(np.int16))
    assert abs(len(read_audio(tmp_path / "r8.wav")) - 8000) <= 1          # resampled to 16 kHz
    with pytest.raises(ValueError):
        read_audio(tmp_path / "x.mp3")


# ------------------------------------------------------------------ N > 1 (gloo, world_size 2)
_WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["REPO"])
from vtc_amd import dist, utils
rank, world, _ = dist.init("gloo")
items = [f"f{i}" for i in range(11)]
mine = utils.shard_list(items, rank, world)
dist.barrier()
slowest = dist.max_over_ranks(1.0 + rank)
total = dist.sum_over_ranks(len(mine))
print(json.dumps({"rank": rank, "mine": mine, "slowest": slowest, "total": total}))
dist.shutdown()
"""


def test_two_rank_sharding_over_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, REPO=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
    assert all(p.returncode == 0 for p in procs)
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["mine"] + outs[1]["mine"] == [f"f{i}" for i in range(11)]
    assert not set(outs[0]["mine"]) & set(outs[1]["mine"])
    assert all(o["slowest"] == 2.0 and o["total"] == 11 for o in outs)


def test_xlsx_manifest(tmp_path):
    """.xlsx manifests (reference src/utils.py:305-312 reads them through polars/openpyxl): a hand-made workbook with shared
    strings, an inline string, numbers, an empty cell and a second sheet that must be ignored."""
    import zipfile
    from vtc_amd import utils
    ns = 'xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main"'
    rel = 'xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"'
    sheet = (f'<worksheet {ns}><sheetData>'
             '<row r="1"><c r="A1" t="s"><v>0</v></c><c r="B1" t="s"><v>1</v></c><c r="C1" t="s"><v>2</v></c></row>'
             '<row r="2"><c r="A2" t="s"><v>3</v></c><c r="B2"><v>12.5</v></c><c r="C2" t="s"><v>5</v></c></row>'
             '<row r="3"><c r="A3" t="inlineStr"><is><t>/data/b.flac</t></is></c><c r="B3"><v>7</v></c></row>'
             '<row r="4"><c r="A4" t="s"><v>4</v></c><c r="C4" t="str"><v>x</v></c></row>'
             '</sheetData></worksheet>')
    other = f'<worksheet {ns}><sheetData><row r="1"><c r="A1" t="inlineStr"><is><t>nope</t></is></c></row></sheetData></worksheet>'
    shared = (f'<sst {ns}><si><t>path</t></si><si><t>duration</t></si><si><t>note</t></si><si><t>/data/a.wav</t></si>'
              '<si><r><t>/data/</t></r><r><t>c.wav</t></r></si><si><t>ok</t></si></sst>')
    wb = (f'<workbook {ns} {rel}><sheets><sheet name="files" sheetId="7" r:id="rId2"/><sheet name="x" sheetId="1" r:id="rId1"/>'
          '</sheets></workbook>')
    rels = ('<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships">'
            '<Relationship Id="rId1" Target="worksheets/sheet1.xml" Type="t"/>'
            '<Relationship Id="rId2" Target="worksheets/sheet2.xml" Type="t"/></Relationships>')
    with zipfile.ZipFile(tmp_path / "m.xlsx", "w") as z:
        z.writestr("xl/workbook.xml", wb)
        z.writestr("xl/_rels/workbook.xml.rels", rels)
        z.writestr("xl/sharedStrings.xml", shared)
        z.writestr("xl/worksheets/sheet1.xml", other)
        z.writestr("xl/worksheets/sheet2.xml", sheet)
    t = utils.load_manifest(tmp_path / "m.xlsx")
    assert t.column_names == ["path", "duration", "note"]
    assert utils.manifest_paths(t) == ["/data/a.wav", "/data/b.flac", "/data/c.wav"]
    assert t.column("duration").to_pylist() == [12.5, 7, None] and t.column("note").to_pylist() == ["ok", None, "x"]
    (tmp_path / "bad.xlsx").write_bytes(b"not a zip")
    with pytest.raises(RuntimeError, match="Failed to read Excel"):
        utils.load_manifest(tmp_path / "bad.xlsx")
    (tmp_path / "old.xls").write_bytes(b"\xd0\xcf\x11\xe0")
    with pytest.raises(RuntimeError, match="Failed to read Excel"):
        utils.load_manifest(tmp_path / "old.xls")
    # malformed workbooks: every failure is the same RuntimeError naming the file
    def book(name, sheet_xml, shared_xml=shared, rels_xml=rels):
        with zipfile.ZipFile(tmp_path / name, "w") as z:
            z.writestr("xl/workbook.xml", wb)
            z.writestr("xl/_rels/workbook.xml.rels", rels_xml)
            z.writestr("xl/sharedStrings.xml", shared_xml)
            z.writestr("xl/worksheets/sheet1.xml", other)
            z.writestr("xl/worksheets/sheet2.xml", sheet_xml)
        return tmp_path / name
    with pytest.raises(RuntimeError, match="Failed to read Excel"):      # shared-string index out of range
        utils.load_manifest(book("idx.xlsx", sheet.replace("<v>5</v>", "<v>99</v>")))
    with pytest.raises(RuntimeError, match="Failed to read Excel"):      # relationship without a Target
        utils.load_manifest(book("rel.xlsx", sheet, rels_xml=rels.replace(' Target="worksheets/sheet2.xml"', "")))
    empty = f'<worksheet {ns}><sheetData><row r="1"/><row r="2"/></sheetData></worksheet>'
    assert utils._read_xlsx(book("empty.xlsx", empty)).num_columns == 0   # rows without cells: an empty table, not a ValueError
    dup = sheet.replace('<c r="C1" t="s"><v>2</v></c>', '<c r="C1" t="s"><v>0</v></c>')       # header: path, duration, path
    t2 = utils._read_xlsx(book("dup.xlsx", dup))
    assert t2.column_names == ["path", "duration", "path_duplicated_0"] and t2.column("path").to_pylist()[0] == "/data/a.wav"


def _biff(typ: int, data: bytes) -> bytes:
    import struct
    return struct.pack("<HH", typ, len(data)) + data


def _xls_workbook_stream() -> bytes:
    """A BIFF8 workbook stream written by hand from [MS-XLS]: globals (BOF, two BOUNDSHEETs, an SST whose third string is
    split over a CONTINUE record that switches from 8-bit to 16-bit characters, EOF) and two sheets; the first sheet holds
    LABELSST, NUMBER, RK (integer, x100 and IEEE forms), MULRK, a FORMULA with a cached string (STRING record) and a
    FORMULA with a cached number."""
    import struct
    def ustr8(t):
        return struct.pack("<HB", len(t), 0) + t.encode("latin-1")
    bof_g = _biff(0x0809, struct.pack("<HHHHII", 0x0600, 0x0005, 0x0DBB, 0x07CC, 0, 6))
    bof_s = _biff(0x0809, struct.pack("<HHHHII", 0x0600, 0x0010, 0x0DBB, 0x07CC, 0, 6))
    eof = _biff(0x000A, b"")
    strings = ["path", "duration", "/data/a.wav", "note", "ok", "/data/\u00e9t\u00e9.wav"]
    # string 2 = "/data/a.wav" is cut after "/data/": the rest arrives in a CONTINUE with a fresh option byte (1 = UTF-16)
    head = struct.pack("<II", len(strings), len(strings)) + ustr8("path") + ustr8("duration") + struct.pack("<HB", 11, 0) + b"/data/"
    cont = b"\x01" + "a.wav".encode("utf-16-le") + ustr8("note") + ustr8("ok") \
        + struct.pack("<HB", len(strings[5]), 1) + strings[5].encode("utf-16-le")
    sst = _biff(0x00FC, head) + _biff(0x003C, cont)
    def cell(r, c):
        return struct.pack("<HHH", r, c, 15)
    rk_int = lambda v: (v << 2) | 2
    rk_cents = lambda v: (v << 2) | 3
    rk_f = lambda x: struct.unpack("<Q", struct.pack("<d", x))[0] >> 32 & 0xFFFFFFFC
    sheet1 = bof_s
    sheet1 += _biff(0x00FD, cell(0, 0) + struct.pack("<I", 0)) + _biff(0x00FD, cell(0, 1) + struct.pack("<I", 1))
    sheet1 += _biff(0x00FD, cell(0, 2) + struct.pack("<I", 3))
    sheet1 += _biff(0x00FD, cell(1, 0) + struct.pack("<I", 2)) + _biff(0x0203, cell(1, 1) + struct.pack("<d", 12.5))
    sheet1 += _biff(0x00FD, cell(1, 2) + struct.pack("<I", 4))
    sheet1 += _biff(0x00FD, cell(2, 0) + struct.pack("<I", 5)) + _biff(0x027E, cell(2, 1) + struct.pack("<I", rk_int(7)))
    sheet1 += _biff(0x0006, cell(3, 0) + b"\x00\x00\x00\x00\x00\x00\xff\xff" + b"\x00" * 8) \
        + _biff(0x0207, struct.pack("<HB", 11, 0) + b"/data/f.wav")
    sheet1 += _biff(0x00BD, struct.pack("<HH", 3, 1) + struct.pack("<HI", 15, rk_cents(1234)) + struct.pack("<HI", 15, rk_f(0.5))
                    + struct.pack("<H", 2))
    sheet1 += _biff(0x0006, cell(4, 1) + struct.pack("<d", 3.0) + b"\x00" * 8)
    sheet1 += eof
    sheet2 = bof_s + _biff(0x0204, cell(0, 0) + ustr8("nope")) + eof
    def globals_(off1, off2):
        bs = lambda off, name: _biff(0x0085, struct.pack("<IBB", off, 0, 0) + struct.pack("<BB", len(name), 0) + name.encode())
        return bof_g + bs(off1, "files") + bs(off2, "x") + sst + eof
    n = len(globals_(0, 0))
    return globals_(n, n + len(sheet1)) + sheet1 + sheet2


def _cfb(stream: bytes, mini: bool) -> bytes:
    """A compound file ([MS-CFB], 512-byte sectors) holding one stream named "Workbook": in regular sectors (padded to the
    4096-byte cut-off) or, for mini=True, in the mini stream of the root entry."""
    import struct
    END, FREE, FATSECT = 0xFFFFFFFE, 0xFFFFFFFF, 0xFFFFFFFD
    def pad(b, n):
        return b + b"\x00" * (-len(b) % n)
    def dirent(name, typ, start, size, child=FREE):
        nm = name.encode("utf-16-le") + b"\x00\x00"
        return nm.ljust(64, b"\x00") + struct.pack("<HBB", len(nm), typ, 1) + struct.pack("<III", FREE, FREE, child) \
            + b"\x00" * 36 + struct.pack("<IQ", start, size)
    sectors, fat = [], []
    def add(data):          # append a chain of sectors, return its first sector id
        data = pad(data, 512)
        first = len(sectors)
        k = len(data) // 512
        for i in range(k):
            sectors.append(data[i * 512:(i + 1) * 512])
            fat.append(first + i + 1 if i + 1 < k else END)
        return first
    if mini:
        assert len(stream) < 4096
        mini_stream = pad(stream, 64)
        n_mini = len(mini_stream) // 64
        minifat = [i + 1 for i in range(n_mini - 1)] + [END]
        minifat_sid = add(struct.pack(f"<{len(minifat)}I", *minifat) + struct.pack("<I", FREE) * (-len(minifat) % 128))
        root_start = add(mini_stream)
        directory = dirent("Root Entry", 5, root_start, len(mini_stream), child=1) + dirent("Workbook", 2, 0, len(stream))
        n_minifat_sec = 1
    else:
        body = stream + b"\x00" * max(0, 4096 - len(stream))
        wb_start = add(body)
        directory = dirent("Root Entry", 5, END, 0, child=1) + dirent("Workbook", 2, wb_start, len(body))
        minifat_sid, n_minifat_sec = END, 0
    dir_sid = add(pad(directory, 512))
    fat_sid = len(sectors)
    fat.append(FATSECT)
    sectors.append(pad(struct.pack(f"<{len(fat)}I", *fat), 512).replace(b"", b"", 0))
    sectors[-1] = struct.pack(f"<{len(fat)}I", *fat) + struct.pack("<I", FREE) * (128 - len(fat))
    header = bytes.fromhex("D0CF11E0A1B11AE1") + b"\x00" * 16 + struct.pack("<HHHHH", 0x3E, 3, 0xFFFE, 9, 6) + b"\x00" * 6 \
        + struct.pack("<IIIIIIIII", 0, 1, dir_sid, 0, 4096, minifat_sid, n_minifat_sec, END, 0) \
        + struct.pack("<I", fat_sid) + struct.pack("<I", FREE) * 108
    assert len(header) == 512 and len(fat) <= 128
    return header + b"".join(sectors)


def test_xls_manifest(tmp_path):
    """Legacy .xls manifests (reference src/utils.py:305-312 reads them through polars -> xlrd).  The workbook is written by
    hand from [MS-CFB] / [MS-XLS] (no Excel-written file is available offline: the reader is unpinned against real Excel
    output), once in regular sectors and once in the mini stream."""
    from vtc_amd import utils
    stream = _xls_workbook_stream()
    for mini in (False, True):
        p = tmp_path / f"m{int(mini)}.xls"
        p.write_bytes(_cfb(stream, mini))
        t = utils.load_manifest(p)
        assert t.column_names == ["path", "duration", "note"]
        assert utils.manifest_paths(t) == ["/data/a.wav", "/data/\u00e9t\u00e9.wav", "/data/f.wav"]
        assert t.column("duration").to_pylist() == [12.5, 7, 12.34, 3] and t.column("note").to_pylist() == ["ok", None, "0.5", None]
    bad = bytearray(_cfb(stream, False))
    bad[0x30:0x34] = b"\x77\x77\x00\x00"                   # directory start beyond the FAT
    (tmp_path / "bad.xls").write_bytes(bytes(bad))
    with pytest.raises(RuntimeError, match="Failed to read Excel"):
        utils.load_manifest(tmp_path / "bad.xls")


def test_reader_task_planning():
    """pipeline/vtc.py::plan_reader_tasks: runs of short files become bundles of at most one batch of windows, 16 MB and 256
    files; long, missing (size 0) and isolated files are tasks of their own; every file appears exactly once, in order."""
    from vtc_amd.pipeline import vtc as cli
    step, batch = 199 * 320, 128
    sec = lambda s: 44 + int(s * 32000)                       # bytes of a mono 16 kHz PCM16 WAV of s seconds
    sizes = [sec(3600)] + [sec(5)] * 100 + [0] + [sec(8)] * 3 + [sec(600)] + [sec(2)] + [sec(1800), sec(1)]
    tasks = cli.plan_reader_tasks(sizes, step, batch)
    assert [i for t in tasks for i in t] == list(range(len(sizes)))                       # a partition, in order
    assert tasks[0] == [0]                                                                # the hour-long file alone
    windows = lambda n: max(1, -(-(n // 2 - 400 + 1) // step))
    for t in tasks:
        if len(t) > 1:
            assert all(0 < sizes[i] <= cli.BUNDLE_FILE_BYTES for i in t)
            assert sum(windows(sizes[i]) for i in t) <= batch and sum(sizes[i] for i in t) <= cli.BUNDLE_BYTES and len(t) <= cli.BUNDLE_FILES
    assert tasks[1] == list(range(1, 65)) and tasks[2] == list(range(65, 101))            # 2 windows per 5 s file: 64 files fill a batch
    assert [101] in tasks and [102, 103, 104] in tasks and [105] in tasks                 # missing file; a short run; the 10-minute file
    assert tasks[-3:] == [[106], [107], [108]]                                            # isolated short files between long ones
    # many tiny files: capped by the file count, then by bytes
    tiny = cli.plan_reader_tasks([sec(0.1)] * 600, step, batch)
    assert [len(t) for t in tiny] == [128, 128, 128, 128, 88]                             # one window each: the batch caps first
    big = cli.plan_reader_tasks([sec(120)] * 10, step, 4096)
    assert [len(t) for t in big] == [4, 4, 2]                                             # 3.8 MB each: 16 MB per bundle
    assert cli.plan_reader_tasks(sizes, step, batch, bundle=False) == [[i] for i in range(len(sizes))]


def test_window_packing_plan_and_stitching():
    """pipeline/vtc.py::pack_forward with a stand-in engine on the CPU (the plan and the stitching are host logic): the
    stand-in writes, for every kept frame of every window, the first sample of that frame's receptive field.  Whatever
    way recordings are cut over forwards, a recording's stitched rows must then be wav[g * 320] for g < frames(n) --
    i.e. the window plan is vtc_infer_waveform's (step T * 320, zero-padded tail, frames inside the audio), nothing is
    lost or doubled at a cut, and every forward but the last is full."""
    import torch
    from collections import deque
    from vtc_amd.pipeline import vtc as cli

    class Stub:
        device, n_labels, window_samples, frames_per_window = torch.device("cpu"), 2, 64000, 199
        step = 199 * 320
        calls = []

        def forward_windows(self, wav, start, valid, dst, keep, logits):
            assert wav.dim() == 1 and wav.is_contiguous()
            assert len(start) <= 8 and int((start + valid).max()) <= wav.numel()
            self.calls.append(len(start))
            for s, v, d, k in zip(start, valid, dst, keep):
                assert 0 < k <= 199 and 400 <= v <= 64000 and (k - 1) * 320 + 400 <= v
                for lab in range(2):
                    logits[d:d + k, lab] = wav[s:s + k * 320:320] + lab

    eng, batch = Stub(), 8
    rng = np.random.default_rng(3)
    lens = [64000 * 3 + 7, 400, 63680 + 399, 63680 + 400, 64000, 5000, 64000 * 11 + 12345, 719, 64000 * 20, 4800, 640000]
    wavs = [torch.from_numpy(rng.standard_normal(n).astype(np.float32)) for n in lens]
    carry, queued, got = deque(), 0, {}
    for i, w in enumerate(wavs):
        n_win = -(-(w.numel() - 400 + 1) // eng.step)
        carry.append(cli.OpenRecording(i, w, n_win))
        queued += n_win
        while queued >= batch or (i == len(wavs) - 1 and queued):
            nw = min(batch, queued)
            done, out = cli.pack_forward(eng, carry, nw)
            queued -= nw
            if done:
                logits, rows, n_samples = cli.packed_decode_input(done, out)
                assert n_samples == [o.n for o in done] and rows[-1] == logits.shape[0]
                for k, o in enumerate(done):
                    got[o.i] = logits[int(rows[k]):int(rows[k + 1])].clone()
    assert not carry and sorted(got) == list(range(len(wavs)))
    assert all(c == batch for c in eng.calls[:-1]) and sum(eng.calls) == sum(-(-(n - 399) // eng.step) for n in lens)
    for i, w in enumerate(wavs):
        frames = (w.numel() - 400) // 320 + 1
        assert got[i].shape == (frames, 2), i
        want = w[:frames * 320:320]
        assert torch.equal(got[i][:, 0], want) and torch.equal(got[i][:, 1], want + 1), i


def test_batch_size_recommendation():
    """reference src/pipeline/resources.py:180-196 for 4 s windows; scaled by the window length for other configs."""
    from vtc_amd.pipeline.vtc import recommend_vtc_batch_size as rec
    assert [rec(v) for v in (288, 80, 79, 48, 40, 32, 24, 16, 8)] == [512, 512, 256, 256, 192, 192, 128, 64, 32]
    assert rec(288, 10.0) == 208 and rec(288, 4.0) == 512 and rec(24, 10.0) == 48 and rec(8, 30.0) == 16
    # trimmed to whole rounds of 256-row GEMM tiles over 256 CUs (never above the ladder's value, never below 80 % of it)
    from vtc_amd.pipeline.vtc import tile_friendly_batch as fit
    assert fit(512, 4.0) == 491 and fit(208, 10.0) == 196 and fit(64, 10.0) == 64 and fit(128, 4.0) == 128
    for cap, chunk in ((512, 4.0), (256, 4.0), (192, 4.0), (208, 10.0), (96, 15.0), (7, 4.0), (33, 2.0)):
        assert max(1, int(cap * 0.8)) <= fit(cap, chunk) <= cap
    # 491 windows of 4 s: 97,709 frames = 382 row tiles; x 4 column tiles = 1,528 tiles = 5.97 rounds, x 12 = 17.9 rounds
    assert -(-491 * 199 // 256) == 382


def test_tile_round_model_agrees_with_the_measured_batch_sweep():
    """packing.py::forward_cost_per_window (rounds of GEMM tiles per window) against the hot-path rates measured on the MI355X at
    the same batch sizes (profiles/r03_batch_friendly.json, tools/batch_sweep.py): wherever the model says one batch is at least
    4 % cheaper per window than another, the measurement must agree on the direction, and the batches the model calls within
    1 % of each other were measured within 2.5 %."""
    from vtc_amd.packing import forward_cost_per_window as cost
    pts = json.loads((Path(__file__).parent.parent / "profiles" / "r03_batch_friendly.json").read_text())["points"]
    for chunk in (4.0, 10.0):
        sel = [(p["batch"], p["audio_hours_per_s"]) for p in pts if p["chunk_duration_s"] == chunk]
        assert len(sel) >= 8
        checked = 0
        for b1, r1 in sel:
            for b2, r2 in sel:
                c1, c2 = cost(b1, chunk), cost(b2, chunk)
                if c1 < 0.96 * c2:
                    assert r1 > r2, (chunk, b1, b2, r1, r2)
                    checked += 1
                elif abs(c1 / c2 - 1) < 0.01:
                    assert abs(r1 / r2 - 1) < 0.025, (chunk, b1, b2, r1, r2)
        assert checked >= 3, (chunk, checked)


def test_launcher_cpu_slices_and_thread_budget():
    """Eight workers share a node: each gets an equal contiguous CPU slice and pools sized from it (launcher docstring)."""
    from vtc_amd.launcher import thread_budget, worker_cpu_slices
    sl = worker_cpu_slices(8, cpus=range(192))
    assert [len(x) for x in sl] == [24] * 8 and sl[0][0] == 0 and sl[7][-1] == 191
    assert sorted(c for x in sl for c in x) == list(range(192))                  # a partition
    assert worker_cpu_slices(3, cpus=[4, 5, 6, 7, 8, 9, 10]) == [[4, 5], [6, 7], [8, 9]]
    assert worker_cpu_slices(4, cpus=[0, 1]) == [[0], [1], [0], [1]]             # fewer CPUs than workers: overlap, never empty
    assert worker_cpu_slices(2, per_worker=3, cpus=range(8)) == [[0, 1, 2], [3, 4, 5]]
    assert thread_budget(24) == {"readers": 16, "intra_op": 1, "flac_threads": 4}
    assert thread_budget(8) == {"readers": 6, "intra_op": 1, "flac_threads": 4}
    assert thread_budget(1) == {"readers": 2, "intra_op": 1, "flac_threads": 1}


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two rank processes by itself (the parent
    never touches the GPU) and rank 0 must report the size of the initialised group.  --spawn-check runs the launch
    plumbing (rendezvous on 127.0.0.1, barrier, max over ranks) over gloo with no GPU work."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                        # only rank 0 prints
    assert lines[0] == {"spawn_check": True, "n_gpus": 2, "max_over_ranks": 2.0, "spawned_by_bench": True}
    # under an external launcher (WORLD_SIZE set) the script is a rank and starts nothing
    r1 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--spawn-check"],
                        env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and json.loads(r1.stdout.strip().splitlines()[-1])["spawned_by_bench"] is False
    # a failing rank ends the run with a non-zero status instead of leaving the others at the barrier
    r2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dtype", "nonsense"], env=env,
                        capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0


def test_host_resampler_matches_oracle_restatement_of_torchaudio():
    """audio.sinc_resample (host form of the GPU front-end) vs oracle.resample_sinc_hann (torchaudio's published
    algorithm restated with torch ops); lengths are ceil(n * 16000 / sr)."""
    from vtc_amd import audio
    rng = np.random.default_rng(3)
    for sr in (8000, 11025, 22050, 32000, 44100, 48000, 96000):
        x = (rng.standard_normal(sr // 3 + 29) * 0.1).astype(np.float32)
        a, b = O.resample_sinc_hann(x, sr), audio.sinc_resample(x, sr)
        assert a.shape == b.shape == (-(-16000 * len(x) // sr),)
        assert np.abs(a - b).max() <= 1e-6, sr
    t = np.arange(48000) / 48000.0
    y = audio.sinc_resample(np.sin(2 * np.pi * 440 * t).astype(np.float32), 48000)
    assert np.abs(y[200:-200] - np.sin(2 * np.pi * 440 * np.arange(len(y)) / 16000.0)[200:-200]).max() < 1e-3
    assert np.array_equal(audio.sinc_resample(np.ones(5, np.float32), 16000), np.ones(5, np.float32))


def test_clip_tar_iterator_and_wav_decode(tmp_path):
    """Host side of the clips caller (no GPU): WebDataset grouping of tar members by key, in-memory WAV decode,
    the errors the caller turns into skipped clips."""
    import io, tarfile
    from scipy.io import wavfile
    from oracle import flac_codec as FL
    from vtc_amd.analysis.vtc_on_clips import decode_clip, iter_tar_samples

    def add(tf, name, data):
        info = tarfile.TarInfo(name)
        info.size = len(data)
        tf.addfile(info, io.BytesIO(data))

    buf = io.BytesIO()
    x = (np.sin(np.arange(8000) / 10.0) * 20000).astype(np.int16)
    wavfile.write(buf, 16000, x)
    with tarfile.open(tmp_path / "s.tar", "w") as tf:
        add(tf, "a_0000.wav", buf.getvalue())
        add(tf, "a_0000.json", b"{}")
        add(tf, "dir/b_0001.seg.json", b"{}")
        add(tf, "dir/b_0001.flac", FL.encode(x.astype(np.int64)[::-1], 16000))
        add(tf, "c_0002.txt", b"nothing")
        add(tf, "d_0003.flac", b"fLaC")
    samples = list(iter_tar_samples(tmp_path / "s.tar"))
    assert [s_["__key__"] for s_ in samples] == ["a_0000", "dir/b_0001", "c_0002", "d_0003"]
    assert set(samples[0]) == {"__key__", "wav", "json"} and set(samples[1]) == {"__key__", "seg.json", "flac"}
    pcm, sr = decode_clip(samples[0])
    assert sr == 16000 and pcm.dtype == np.int16 and np.array_equal(pcm, x) and pcm.flags.writeable
    pcm, sr = decode_clip(samples[1])                       # flac member: the library's decoder
    assert sr == 16000 and pcm.dtype == np.int16 and np.array_equal(pcm, x[::-1]) and pcm.flags.writeable
    with pytest.raises(ValueError, match="no audio"):
        decode_clip(samples[2])
    with pytest.raises(RuntimeError, match="FLAC"):
        decode_clip(samples[3])


def test_fast_wav_reader_matches_scipy(tmp_path):
    """audio.read_pcm's own RIFF parser (PCM16 / float32, mono and multi-channel, extra chunks, odd chunk sizes) against
    scipy.io.wavfile; other sample formats fall back to scipy."""
    import struct
    from scipy.io import wavfile
    from vtc_amd import audio
    rng = np.random.default_rng(9)
    cases = {"m16.wav": (16000, (rng.standard_normal(5001) * 8000).astype(np.int16)),
             "s16.wav": (44100, (rng.standard_normal((777, 2)) * 8000).astype(np.int16)),
             "f32.wav": (48000, rng.standard_normal((300, 3)).astype(np.float32) * 0.1),
             "u8.wav": (8000, rng.integers(0, 255, 400).astype(np.uint8)),
             "i32.wav": (16000, (rng.standard_normal(400) * 1e8).astype(np.int32))}
    for name, (sr, x) in cases.items():
        wavfile.write(tmp_path / name, sr, x)
        got, gsr = audio.read_pcm(tmp_path / name)
        assert gsr == sr
        if x.dtype in (np.int16, np.float32):
            assert audio._fast_wav(tmp_path / name) is not None
            assert got.dtype == x.dtype and np.array_equal(got, x)
        else:
            assert audio._fast_wav(tmp_path / name) is None and got.dtype == np.float32
    # 24-bit PCM (what field recorders write): the reader hands back the packed bytes, uint8 (n, channels, 3), for the GPU
    # front-end to unpack; the host route (pcm24_to_float32) gives exactly what scipy's left-justified int32 / 2^31 gives
    for name, ch, sr in (("m24.wav", 1, 48000), ("s24.wav", 2, 44100)):
        v = rng.integers(-(1 << 23), 1 << 23, (601, ch)).astype(np.int32)
        v[0], v[1] = (1 << 23) - 1, -(1 << 23)
        b = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=-1).astype(np.uint8)
        data = b.tobytes()
        hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, sr, sr * ch * 3, ch * 3, 24) + \
            b"data" + struct.pack("<I", len(data))
        (tmp_path / name).write_bytes(hdr + data)
        got, gsr = audio.read_pcm(tmp_path / name)
        assert gsr == sr and got.dtype == np.uint8 and got.shape == (601, ch, 3) and np.array_equal(got, b)
        ssr, sx = wavfile.read(tmp_path / name)
        assert ssr == sr and np.array_equal(audio.pcm24_to_float32(got).reshape(601, ch), audio.to_float32(sx).reshape(601, ch))
        want = audio.to_float32(sx).reshape(601, ch).mean(axis=1, dtype=np.float32)
        assert np.array_equal(audio.read_audio(tmp_path / name), audio.sinc_resample(want, sr))
    # a LIST chunk of odd length before the data chunk, and a data size field larger than the file (truncated recording)
    raw = (tmp_path / "m16.wav").read_bytes()
    i = raw.index(b"data")
    odd = raw[:i] + b"LIST" + struct.pack("<I", 5) + b"abcde\0" + b"data" + struct.pack("<I", 0xFFFFFFF0) + raw[i + 8:]
    (tmp_path / "odd.wav").write_bytes(odd)
    got, gsr = audio.read_pcm(tmp_path / "odd.wav")
    assert gsr == 16000 and np.array_equal(got, cases["m16.wav"][1])
    (tmp_path / "junk.wav").write_bytes(b"not a wav file at all")
    with pytest.raises(Exception):
        audio.read_pcm(tmp_path / "junk.wav")


def test_g711_and_rf64_wav_files(tmp_path):
    """WAV sub-formats beyond PCM / float: 8-bit G.711 A-law and mu-law (format tags 6 and 7: telephone-speech corpora) are
    expanded with the Recommendation's tables, and RF64 / BW64 headers (64-bit sizes in a 'ds64' chunk, for recordings past
    4 GB) are read.  The tables are pinned by G.711's own properties: the extreme and smallest values of the
    Recommendation, and re-compression with the Recommendation's ENCODER (segment search: another algorithm) giving back
    every code (mu-law 0x7F, 'negative zero', re-encodes as 0xFF: the one known exception)."""
    import struct
    from vtc_amd import audio

    ulaw, alaw = audio.g711_table("ulaw"), audio.g711_table("alaw")
    assert (ulaw[0x00], ulaw[0x80], ulaw[0xFF], ulaw[0x7F], ulaw[0xFE], ulaw[0x7E]) == (-32124, 32124, 0, 0, 8, -8)
    assert (alaw[0xD5], alaw[0x55], alaw[0xAA], alaw[0x2A], alaw[0xD4], alaw[0x54]) == (8, -8, 32256, -32256, 24, -24)

    def lin2ulaw(x):
        sign = 0x80 if x < 0 else 0
        x = min(abs(x), 32635) + 0x84
        e = max(x.bit_length() - 8, 0)
        return ~(sign | (e << 4) | ((x >> (e + 3)) & 0x0F)) & 0xFF

    def lin2alaw(x):
        sign = 0x80 if x >= 0 else 0
        x = min(abs(x), 32767)
        e = 0 if x < 256 else x.bit_length() - 8
        m = x >> 4 if x < 256 else (x >> (e + 3)) & 0x0F
        return (sign | (e << 4) | m) ^ 0x55

    assert [c for c in range(256) if lin2ulaw(int(ulaw[c])) != c] == [0x7F]
    assert [c for c in range(256) if lin2alaw(int(alaw[c])) != c] == []
    for t in (ulaw, alaw):        # 255 / 256 distinct levels, odd symmetry
        assert len(set(t.tolist())) >= 255 and sorted(t.tolist()) == sorted((-t.astype(np.int32)).tolist())

    def wav(tag, bits, ch, sr, payload, magic=b"RIFF", extra=b"", data_size=None, ds64=None):
        fmt = struct.pack("<HHIIHH", tag, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits) + (b"\0\0" if tag != 1 else b"")
        body = b"WAVE" + (b"ds64" + struct.pack("<I", 28) + ds64 if ds64 else b"")
        body += b"fmt " + struct.pack("<I", len(fmt)) + fmt
        body += b"data" + struct.pack("<I", len(payload) if data_size is None else data_size) + payload + extra
        return magic + struct.pack("<I", 0xFFFFFFFF if ds64 else len(body)) + body

    rng = np.random.default_rng(21)
    codes = rng.integers(0, 256, 4001 * 2, dtype=np.uint8)
    for name, tag, table in (("a.wav", 6, alaw), ("u.wav", 7, ulaw)):
        (tmp_path / name).write_bytes(wav(tag, 8, 2, 8000, codes.tobytes()))
        got, sr = audio.read_pcm(tmp_path / name)
        assert sr == 8000 and got.dtype == np.int16 and np.array_equal(got, table[codes].reshape(-1, 2))
        mono = audio.read_audio(tmp_path / name)                                  # mix + 8 -> 16 kHz on the host route
        assert mono.dtype == np.float32 and abs(len(mono) - 8002) <= 1 and np.abs(mono).max() <= 1.0
    (tmp_path / "m.wav").write_bytes(wav(7, 8, 1, 16000, codes[:999].tobytes()))
    dst = np.zeros(4096, dtype=np.int16)
    assert audio.read_plain_wav_into(tmp_path / "m.wav", dst, 0) is None          # not PCM16: staged on its own, not bundled
    got, _ = audio.read_pcm(tmp_path / "m.wav")
    assert np.array_equal(got, ulaw[codes[:999]])

    # AIFF-C carries the same two codecs ('ulaw' / 'alaw'; COMM's sampleSize says 16, the data is one byte per sample)
    def aifc_g711(ctype, ch, payload):
        rate80 = struct.pack(">HQ", 16383 + 12, 8000 << (63 - 12))                 # 8000 Hz as an 80-bit extended float
        comm = struct.pack(">hIh", ch, len(payload) // ch, 16) + rate80 + ctype + b"\x00\x00"
        ssnd = struct.pack(">II", 0, 0) + payload
        body = b"AIFC" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
        return b"FORM" + struct.pack(">I", len(body)) + body

    for ctype, table in ((b"ulaw", ulaw), (b"ALAW", alaw)):
        (tmp_path / "g.aifc").write_bytes(aifc_g711(ctype, 2, codes[:2000].tobytes()))
        got, sr = audio.read_pcm(tmp_path / "g.aifc")
        assert sr == 8000 and got.dtype == np.int16 and np.array_equal(got, table[codes[:2000]].reshape(-1, 2))

    x = (rng.standard_normal(3000) * 9000).astype(np.int16)
    ds64 = struct.pack("<QQQI", 0, x.nbytes, len(x), 0)
    junk = b"LIST" + struct.pack("<I", 6) + b"trailr"                              # a chunk BEHIND the data: ds64's size must be used
    for magic in (b"RF64", b"BW64"):
        (tmp_path / "big.wav").write_bytes(wav(1, 16, 1, 16000, x.tobytes(), magic=magic, extra=junk, data_size=0xFFFFFFFF, ds64=ds64))
        got, sr = audio.read_pcm(tmp_path / "big.wav")
        assert sr == 16000 and got.dtype == np.int16 and np.array_equal(got, x)
        assert audio.read_plain_wav_into(tmp_path / "big.wav", dst, 5) == 3000 and np.array_equal(dst[5:3005], x)


def test_ima_adpcm_wav_against_audioop(tmp_path):
    """4-bit IMA / DVI ADPCM in RIFF/WAVE (format tag 0x0011) through vtc_ima_adpcm_decode.  Pinned against an independent
    implementation of the same codec, the standard library's audioop (Intel/DVI ADPCM): every block is encoded with
    audioop.lin2adpcm from the state its header stores, and the library must give back exactly what audioop.adpcm2lin gives
    for that block.  The WAVE layout differs from audioop's stream in two ways that the test spells out: codes sit LOW nibble
    first (audioop: high first), and channels alternate in 4-byte groups of eight codes."""
    audioop = pytest.importorskip("audioop")
    import struct
    from vtc_amd.audio import read_pcm

    swap = bytes(((b & 15) << 4) | (b >> 4) for b in range(256))

    def ima_wav(path, pcm, sr, align):
        n, ch = pcm.shape
        spb = 1 + (align - 4 * ch) // (4 * ch) * 8
        blocks, want, index = [], np.zeros_like(pcm), [0] * ch
        for a in range(0, n, spb):
            part = pcm[a:a + spb]
            if len(part) < spb:
                part = np.concatenate([part, np.zeros((spb - len(part), ch), np.int16)])
            head, codes = b"", []
            for c in range(ch):
                first = int(part[0, c])
                head += struct.pack("<hBB", first, index[c], 0)
                enc, state = audioop.lin2adpcm(part[1:, c].astype("<i2").tobytes(), 2, (first, index[c]))
                dec, _ = audioop.adpcm2lin(enc, 2, (first, index[c]))
                blk = np.concatenate([[first], np.frombuffer(dec, "<i2")])
                want[a:a + spb, c] = blk[:len(pcm[a:a + spb])]
                index[c] = state[1]
                codes.append(enc.translate(swap))
            body = b"".join(codes[c][g * 4:g * 4 + 4] for g in range((spb - 1) // 8) for c in range(ch))
            blocks.append(head + body)
        data = b"".join(blocks)
        fmt = struct.pack("<HHIIHHHH", 0x11, ch, sr, sr * align // spb, align, 4, 2, spb)
        fact = struct.pack("<I", n)
        riff = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact + \
            b"data" + struct.pack("<I", len(data)) + data
        path.write_bytes(b"RIFF" + struct.pack("<I", len(riff)) + riff)
        return want

    rng = np.random.default_rng(12)
    t = np.arange(16000 * 3) / 16000.0
    sig = 0.6 * np.sin(2 * np.pi * 220 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 1.5 * t)) + 0.05 * rng.standard_normal(len(t))
    sig[5000:5200] = 1.2                      # clipping drives the predictor into its clamp
    mono = np.clip(np.round(sig * 32767), -32768, 32767).astype(np.int16)[:, None]
    stereo = np.stack([mono[:, 0], np.roll(mono[:, 0], 777) // 3], axis=1)[: 16000 * 2 + 123]
    for name, pcm, sr, align in (("m256.wav", mono, 16000, 256), ("m1024.wav", mono[:10001], 8000, 1024),
                                 ("s512.wav", stereo, 22050, 512), ("s2048.wav", stereo, 44100, 2048)):
        want = ima_wav(tmp_path / name, pcm, sr, align)
        got, got_sr = read_pcm(tmp_path / name)
        got = got.reshape(len(got), -1)
        assert got_sr == sr and got.dtype == np.int16 and got.shape == pcm.shape, name
        assert np.array_equal(got, want), name
        err = np.abs(got.astype(np.int32) - pcm.astype(np.int32))
        assert np.median(err) < 400, name      # and it is the signal, not just the same codes
    # malformed: a step index above 88 in a block header is an error, not a read past the table
    raw = bytearray((tmp_path / "m256.wav").read_bytes())
    at = raw.index(b"data") + 8
    raw[at + 2] = 120
    (tmp_path / "bad.wav").write_bytes(bytes(raw))
    with pytest.raises(Exception, match="step index"):
        read_pcm(tmp_path / "bad.wav")


def test_damaged_audio_files_raise_and_never_hang(tmp_path):
    """A manifest of thousands of files holds damaged ones.  read_pcm on mutated WAV (PCM16, stereo 44.1 kHz, 8-bit), AIFF and
    FLAC files (bit flips, random bytes, truncation, insertions, 32-bit size fields set to extreme values) must either return
    int16 / float32 samples or raise an Exception (which the CLI turns into an error row, reference vtc.py:245-249), quickly,
    whatever the header claims (a size field of 4 GB must not allocate 4 GB)."""
    import struct
    import time
    import warnings
    from scipy.io import wavfile
    from oracle import flac_codec as FL
    from vtc_amd.audio import read_pcm, write_wav

    x = (0.3 * np.sin(np.arange(8000) * 0.05)).astype(np.float32)
    seeds = {}
    write_wav(tmp_path / "a.wav", x)
    wavfile.write(tmp_path / "b.wav", 44100, np.stack([x, x], 1))
    wavfile.write(tmp_path / "c.wav", 8000, (x * 127 + 128).astype(np.uint8))
    pcm_be = (x * 32767).astype(">i2").tobytes()
    comm = struct.pack(">hIh", 1, len(x), 16) + b"\x40\x0c\xfa\x00\x00\x00\x00\x00\x00\x00"      # 16 kHz as 80-bit extended
    body = b"AIFF" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">III", len(pcm_be) + 8, 0, 0) + pcm_be
    (tmp_path / "d.aiff").write_bytes(b"FORM" + struct.pack(">I", len(body)) + body)
    (tmp_path / "e.flac").write_bytes(FL.encode((x * 32767).astype(np.int64), 16000))
    for n in ("a.wav", "b.wav", "c.wav", "d.aiff", "e.flac"):
        seeds[n] = (tmp_path / n).read_bytes()
        got, sr = read_pcm(tmp_path / n)                          # the seeds themselves read
        assert len(got) == len(x) and sr in (16000, 44100, 8000)
    rng = np.random.default_rng(0)
    outcomes = {"ok": 0}
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(3000):
            name = list(seeds)[it % len(seeds)]
            d = bytearray(seeds[name])
            for _ in range(int(rng.integers(1, 5))):
                at = int(rng.integers(len(d))) if rng.random() < 0.5 else int(rng.integers(min(len(d), 80)))
                op = int(rng.integers(5))
                if op == 0:
                    d[at] ^= 1 << int(rng.integers(8))
                elif op == 1:
                    d[at] = int(rng.integers(256))
                elif op == 2:
                    del d[at:]
                elif op == 3:
                    d[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
                else:
                    d[at:at + 4] = struct.pack("<I", int(rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, len(d) * 3])))
                if not d:
                    d = bytearray(b"\0")
            p = tmp_path / ("m" + Path(name).suffix)
            p.write_bytes(bytes(d))
            try:
                got, sr = read_pcm(p)
                assert got.dtype in (np.int16, np.float32, np.uint8) and got.nbytes < (64 << 20)
                outcomes["ok"] += 1
            except Exception as e:       # anything that is an Exception becomes an error row; a crash or a hang would not get here
                outcomes[type(e).__name__] = outcomes.get(type(e).__name__, 0) + 1
    assert time.time() - t0 < 120 and outcomes["ok"] > 300 and sum(outcomes.values()) == 3000, outcomes


def test_columnar_segments_and_meta_equal_the_scalar_forms():
    """The CLI gathers segments as column blocks (core.intervals.intervals_to_columns, metadata.vtc_meta_row_columns,
    utils.segments_table_from_blocks); they must give exactly the rows of the scalar mirror of the reference
    (intervals_to_segments / vtc_meta_row / segments_table): same doubles, same JSON, same table."""
    from vtc_amd.core.intervals import intervals_to_columns, intervals_to_segments
    from vtc_amd.core.metadata import vtc_meta_row, vtc_meta_row_columns
    rng = np.random.default_rng(3)
    labels = ["KCHI", "OCH", "MAL", "FEM"]
    # every frame index of a 16 h file is a possible bound: sample them densely at both ends and at random in between
    i = np.unique(np.concatenate([np.arange(0, 60000), rng.integers(0, 2_880_000, 200_000), np.arange(2_870_000, 2_880_000)]))
    starts = i * 320
    ends = (i + rng.integers(0, 400, size=len(i))) * 320 + 400
    on, off, dur, keep = intervals_to_columns(starts, ends)
    assert keep.all()
    for k in rng.integers(0, len(i), 20000).tolist() + list(range(3000)):
        a, b = round(int(starts[k]) / 16000, 3), round(int(ends[k]) / 16000, 3)
        assert on[k] == a and off[k] == b and dur[k] == round(b - a, 3), k
    # a degenerate interval (end <= start) is dropped by both forms
    _, _, _, kp = intervals_to_columns([320, 640], [320, 640 + 400])
    assert list(kp) == [False, True]
    assert intervals_to_segments([(320, 320, "FEM")], "u") == []
    # whole files: rows, meta row and table
    blocks, rows_all = [], []
    for f in range(6):
        n = [0, 1, 7, 300, 2500, 40][f]
        lab = np.sort(rng.integers(0, 4, n)) if f != 4 else rng.integers(0, 4, n)      # decoder order is label-major
        s = rng.integers(0, 100000, n) * 320
        e = s + rng.integers(0, 300, n) * 320 + 400
        segs = intervals_to_segments([(int(a), int(b), labels[int(c)]) for a, b, c in zip(s, e, lab)], f"file{f}")
        o1, o2, d, kp = intervals_to_columns(s, e)
        block = (f"file{f}", o1[kp], o2[kp], d[kp], lab[kp].astype(np.int32))
        assert vtc_meta_row_columns(f"file{f}", 0.5, block[3], block[4], labels, 0.9, 0.1) == vtc_meta_row(f"file{f}", 0.5, segs, 0.9, 0.1)
        blocks.append(block)
        rows_all.extend(segs)
    got = utils.segments_table_from_blocks(blocks, labels)
    assert got.equals(utils.segments_table(rows_all)) and got.schema == utils.SEGMENT_SCHEMA
    assert utils.segments_table_from_blocks([], labels).num_rows == 0
    assert utils.segments_table_from_blocks(blocks[:1], labels).num_rows == 0


def test_launcher_builds_one_worker_per_device(monkeypatch):
    """vtc_amd.launcher: options before "--" are the launcher's, everything after goes to every worker; worker i sees one
    device and gets --array_id i --array_count N (the reference's SLURM array, slurm/vtc.slurm:10)."""
    from vtc_amd import launcher
    calls, envs = [], []

    class FakeProc:
        def __init__(self, cmd, env):
            calls.append((cmd, env["HIP_VISIBLE_DEVICES"], env["HSA_ENABLE_IPC_MODE_LEGACY"]))
            envs.append(env)

        def wait(self):
            return 0 if len(calls) != 99 else 1

    monkeypatch.setattr(launcher.subprocess, "Popen", lambda cmd, env: FakeProc(cmd, env))
    assert launcher.main(["chunks30", "--gpus", "3", "--", "--checkpoint", "best.ckpt", "--balance"]) == 0
    assert [c[1] for c in calls] == ["0", "1", "2"] and all(c[2] == "0" for c in calls)
    slices = launcher.worker_cpu_slices(3)
    for i, (cmd, _, _) in enumerate(calls):
        readers = str(launcher.thread_budget(len(slices[i]))["readers"])
        assert cmd[1:] == ["-m", "vtc_amd.pipeline.vtc", "chunks30", "--array_id", str(i), "--array_count", "3",
                           "--prefetch", readers, "--checkpoint", "best.ckpt", "--balance"]
        assert envs[i]["VTC_WORKER_CPUS"] == ",".join(str(c) for c in slices[i]) and envs[i]["OMP_NUM_THREADS"] == "1"
    calls.clear(); envs.clear()
    assert launcher.main(["ds", "--devices", "0,0", "--no_affinity", "--", "--batch_size", "8", "--prefetch", "3"]) == 0   # two workers on one GPU
    assert [c[1] for c in calls] == ["0", "0"] and calls[1][0][-6:] == ["--array_count", "2", "--batch_size", "8", "--prefetch", "3"]
    calls.clear(); envs.clear()
    assert launcher.main(["ds", "--gpus", "2", "--per_gpu", "2", "--no_affinity"]) == 0                                   # 2 GPUs x 2 workers
    assert [c[1] for c in calls] == ["0", "0", "1", "1"] and [c[0][c[0].index("--array_id") + 1] for c in calls] == ["0", "1", "2", "3"]
    assert all(c[0][c[0].index("--array_count") + 1] == "4" for c in calls)
    assert "VTC_WORKER_CPUS" not in envs[0] or envs[0].get("VTC_WORKER_CPUS") == os.environ.get("VTC_WORKER_CPUS")
    assert launcher.main(["ds", "--gpus", "3", "--devices", "0,1"]) == 2                       # count mismatch
