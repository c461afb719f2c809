"""CPU: the oracle against the committed golden vectors (tests/golden/, made by make_golden.py)."""

import json
import math

import numpy as np
import pytest
import torch

from oracle import vtc_oracle as O
from vtc_amd import synth


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.loads((golden_dir / "golden_kat.json").read_text())


def test_geometry_kat(kat):
    g = kat["geometry"]
    for n, frames in g["n_frames"].items():
        assert O.conv_out_len(int(n)) == frames
        assert O.n_frames_for_samples(int(n)) == frames
    assert O.chunk_artifact_n_windows(64000) == g["N_WINDOWS_4s"]
    assert O.plan_windows(10 * 64000, 4.0).step == g["CHUNK_STEP_F"]
    assert (199 - 1) * O.RF_STEP + O.RF_SIZE == g["FULL_CHUNK_DUR_F"]
    for n, sec in g["peak_seconds"].items():
        assert O.peak_seconds(int(n)) == pytest.approx(sec, abs=1e-12)
    assert O.RF_STEP == 320 and O.RF_SIZE == 400


def test_closed_form_equals_layerwise():
    for n in list(range(380, 2000, 7)) + [64000, 63999, 160000, 921_600_000]:
        assert O.conv_out_len(n) == O.n_frames_for_samples(n)


@pytest.mark.parametrize("n", [400, 719, 720, 64000, 64001, 63680 + 399, 63680 + 400, 3 * 63680 + 12345, 10 * 63680])
def test_window_plan_covers_every_frame_once(n):
    plan = O.plan_windows(n, 4.0)
    assert plan.total_frames == O.n_frames_for_samples(n)
    assert all(s == i * plan.step for i, s in enumerate(plan.starts))
    assert all(k == plan.frames_per_window for k in plan.keep[:-1])
    assert all(0 < v <= plan.window for v in plan.valid)


def test_window_plan_too_short():
    assert O.plan_windows(399, 4.0).starts == []
    with pytest.raises(ValueError):
        O.apply_model_on_waveform({}, np.zeros(399, np.float32), 4.0)


def test_encoder_golden(golden_dir):
    info = json.loads((golden_dir / "golden_encoder.json").read_text())
    assert info["hf_hubert_max_abs_diff_last_hidden"] is not None and info["hf_hubert_max_abs_diff_last_hidden"] <= 1e-5
    z = np.load(golden_dir / "encoder_1s.npz")
    w = O.make_weights(seed=info["seed_weights"])
    rng = np.random.default_rng(info["seed_input"])
    wav = rng.standard_normal((2, 16000), dtype=np.float32) * np.float32(0.1)
    wav[1, 9000:] = 0.0
    with torch.no_grad():
        st = O.encoder_stages(w, torch.from_numpy(wav))
    np.testing.assert_allclose(st["logits"].numpy(), z["logits"], atol=2e-5, rtol=0)
    for k, v in st.items():
        if k == "logits":
            continue
        a = v.numpy()
        np.testing.assert_allclose(a[:, :8, :8], z[k + "_slice"], atol=2e-5, rtol=0)
        s = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
        np.testing.assert_allclose(s, z[k + "_sum"], rtol=1e-5, atol=1e-2)


def test_stitching_is_concatenation_of_window_frames():
    """Frame g of the stitched output is frame g % T of window g // T (chunk_artifact.py:71-73)."""
    w = O.make_weights(seed=3)
    rng = np.random.default_rng(5)
    chunk = 0.5
    T = O.conv_out_len(8000)
    wav = rng.standard_normal(2 * T * 320 + 3000, dtype=np.float32) * 0.1
    full = O.apply_model_on_waveform(w, wav, chunk, batch_size=2)
    assert full.shape[0] == O.n_frames_for_samples(len(wav))
    second = np.zeros(8000, np.float32)
    second[:] = wav[T * 320: T * 320 + 8000]
    lg = O.encoder_logits(w, torch.from_numpy(second[None]))[0]
    torch.testing.assert_close(full[T: 2 * T], lg, atol=1e-5, rtol=0)


def test_decode_kat(kat):
    for case in kat["decode"]:
        got = O.create_intervals(np.array(case["mask"]), labels=case["labels"])
        assert got == [tuple(x) for x in case["intervals"]]


def test_thresholds_inclusive_and_bounds():
    x = torch.tensor([[0.0, -1e-3, 20.0, 1e-3]])
    thr = {l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in O.DEFAULT_LABELS}
    assert O.apply_thresholds(x, thr).tolist() == [[1, 0, 1, 1]]
    thr["KCHI"] = {"lower_bound": 0.5, "upper_bound": 0.9}
    assert O.apply_thresholds(x, thr).tolist() == [[1, 0, 0, 1]]


def test_segments_kat(kat):
    for case in kat["segments"]:
        assert O.intervals_to_segments([tuple(i) for i in case["intervals"]], case["uid"]) == case["rows"]


def test_meta_rows_match_reference_module(golden_dir):
    g = json.loads((golden_dir / "golden_meta.json").read_text())
    for case in g["meta_rows"]:
        assert O.vtc_meta_row(*case["args"]) == case["row"]
    for case in g["error_rows"]:
        row = O.vtc_error_row(*case["args"])
        for k, v in case["row"].items():
            if v is None:
                assert math.isnan(row[k])
            else:
                assert row[k] == v
        assert set(row) == set(case["row"])


def test_merge_kat(kat):
    m = kat["merge"]
    rows = [dict(uid=u, label=l, onset=a, offset=b, duration=round(b - a, 3)) for u, l, a, b in m["rows"]]
    got = O.merge_segments(rows, m["collar"], m["min_on"])
    assert [[r["uid"], r["label"], r["onset"], r["offset"], r["duration"]] for r in got] == m["expected"]


def test_shard_kat(kat):
    s = kat["shard"]
    items = list(range(s["n"]))
    assert [O.shard_list(items, i, s["count"]) for i in range(s["count"])] == s["slices"]


def test_sigmoid_summary():
    lg = torch.tensor([[0.0, 2.0], [-2.0, 0.0]])
    mx, mean = O.sigmoid_summary(lg)
    assert mx == round(float(torch.sigmoid(torch.tensor(2.0))), 4)
    assert mean == 0.5


def test_head_calibration_fixture_matches_the_generator(golden_dir):
    doc = json.loads((golden_dir / "head_calibration.json").read_text())
    assert doc["target_logit_std"] == 2.0
    for key, e in doc["entries"].items():
        assert synth.HEAD_CALIBRATION[key] == e["uncalibrated_logit_std"]
        fam, seed = key.split(":")
        a = synth.make_weights(seed=int(seed), family=fam)
        b = synth.make_weights(seed=int(seed), family=fam, calibrated=True)
        scale = np.float32(2.0 / e["uncalibrated_logit_std"])
        bias = a["head.bias"]
        if "label_mean_logit" in e:            # family "trained": the head is centred before it is scaled
            assert synth.HEAD_CENTER[key] == e["label_mean_logit"]
            bias = bias - np.asarray(e["label_mean_logit"], dtype=np.float32)
        assert np.array_equal(b["head.weight"], a["head.weight"] * scale) and np.array_equal(b["head.bias"], bias * scale)


def test_trained_like_family_has_outlier_channels_and_a_balanced_head():
    """synth.make_weights(family="trained"): six channels of every LayerNorm output sit at 50-100 x the rest, LayerNorm gains
    are far from 1, and the calibrated head crosses the threshold on every label (oracle, 2 windows of 4 s)."""
    w = O.make_weights(seed=0, family="trained", calibrated=True)
    base = O.make_weights(seed=0, family="hubert")
    assert torch.equal(w["feature_extractor.conv_layers.3.conv.weight"], base["feature_extractor.conv_layers.3.conv.weight"])
    g = w["encoder.layers.5.final_layer_norm.weight"]
    assert float(g.max()) > 5.0 and float(g.min()) <= 0.1 + 1e-6
    audio = synth.synth_audio(8.0, seed=9100)
    x = torch.from_numpy(np.stack([audio[:64000], audio[64000:128000]]))
    ranges = {}
    with torch.no_grad():
        st = O.encoder_stages(w, x, ranges=ranges)
    for k in ("enc_in", "layer0", "layer11"):
        a = st[k].abs().amax(dim=(0, 1))
        top = torch.sort(a, descending=True).values
        rms = st[k].pow(2).mean(dim=(0, 1)).sqrt().median()
        assert 50.0 <= float(top[5]) and float(top[0]) <= 130.0, (k, top[:8])
        assert float(top[5]) / float(rms) >= 20.0, (k, float(top[5]), float(rms))
    assert 50.0 <= ranges["pre_layernorm_rows"] < 65504.0 / 16
    lg = st["logits"].reshape(-1, 4)
    frac = (lg > 0).float().mean(dim=0)
    assert float(frac.min()) > 0.1 and float(frac.max()) < 0.9, frac


def test_boundary_attribution_accounts_for_every_mismatch():
    """oracle/parity_metrics.py::attribute on synthetic logit tracks: with small noise every unmatched segment (raw and
    merged, both directions) is traced to a flipped frame-label in its range; a decode-like difference that is NOT a
    flip near the threshold (a run cut by hand where the oracle logit is far from 0) shows up as such."""
    from oracle.parity_metrics import compare
    g = torch.Generator().manual_seed(0)
    n = 20000
    base = torch.cumsum(torch.randn(n, 4, generator=g) * 0.15, 0)
    base = base - torch.nn.functional.avg_pool1d(base.T[None], 501, 1, 250, count_include_pad=False)[0].T
    got = base + torch.randn(n, 4, generator=g) * 0.004
    c = compare(base, got)
    a = c["attribution"]
    assert c["flipped_frame_labels"] > 20 and a["raw"]["unmatched"] > 10 and a["merged"]["unmatched"] > 0
    assert a["raw"]["unattributed"] == 0 and a["merged"]["unattributed"] == 0
    assert a["max_oracle_abs_logit_at_flips"] <= c["max_abs_logit_err"] <= 0.03
    # identical logits: nothing to attribute
    z = compare(base, base.clone())["attribution"]
    assert z["flipped_frame_labels"] == 0 and z["raw"]["unmatched"] == 0 and z["merged"]["unmatched"] == 0
    # a gross difference: the flip is attributed, but its oracle logit is far from the threshold and the test bars catch it
    k = int(torch.argmax(base[:, 0]))
    bad = base.clone()
    bad[k, 0] = -1.0
    b = compare(base, bad)["attribution"]
    assert b["raw"]["unattributed"] == 0 and b["max_oracle_abs_logit_at_flips"] == pytest.approx(float(base[k, 0]))
    assert b["max_oracle_abs_logit_at_flips"] > 0.5
