"""GPU parity tests: the HIP path (through the C ABI, libvtc_amd.so) against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full size (10 s windows,
batch 64) -- through size-independent properties.

Tolerances are BASELINE.json north_star's: per-frame logits within 2e-2 abs, frame-label agreement >= 99.9 %,
integer/index work (tiling, stitching, threshold decode) bit-exact.
"""

import json
import math
from pathlib import Path

import numpy as np
import pyarrow.parquet as pq
import pytest
import torch

from oracle import vtc_oracle as O
from vtc_amd import lib, synth
from vtc_amd.segma_compat.inference import logit_bounds

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-2          # north_star: per-frame logits within 2e-2 abs
AGREEMENT = 0.999         # north_star: per-frame label agreement >= 99.9 %
THR = {l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in O.DEFAULT_LABELS}


@pytest.fixture(scope="module")
def weights():
    return O.make_weights(seed=0)


def _engine(weights, chunk, batch, dtype="f16"):
    from vtc_amd.engine import VtcEngine
    return VtcEngine(weights, n_labels=4, chunk_duration_s=chunk, max_batch=batch, mfma_dtype=dtype, device=0)


def _conv_lengths(w):
    out = []
    for k, s in zip(O.CONV_KERNELS, O.CONV_STRIDES):
        w = (w - k) // s + 1
        out.append(w)
    return out


# ----------------------------------------------------------------------------- encoder stages
@pytest.mark.parametrize("dtype,conv_tol,hid_tol", [("f16", 1e-2, 1.5e-2), ("bf16", 6e-2, 1.2e-1)])
def test_every_stage_against_oracle(weights, dtype, conv_tol, hid_tol):
    chunk, B = 1.0, 3
    W = 16000
    T = O.conv_out_len(W)
    rng = np.random.default_rng(1)
    wav = rng.standard_normal((B, W), dtype=np.float32) * np.float32(0.1)
    wav[-1, W // 2:] = 0
    with torch.no_grad():
        st = O.encoder_stages(weights, torch.from_numpy(wav))
    eng = _engine(weights, chunk, B, dtype)
    arena = torch.from_numpy(wav.reshape(-1).copy()).cuda()
    start, valid = np.arange(B) * W, np.full(B, W)
    valid[-1] = W // 2          # the tail of the last window is zero padding supplied by the kernel
    dst, keep = np.arange(B) * T, np.full(B, T)
    logits = torch.zeros((B * T, 4), device="cuda")
    names = [f"conv{i}" for i in range(7)] + ["proj", "enc_in"] + [f"layer{l}" for l in range(12)]
    for stage, name in enumerate(names):
        eng.debug_stop_after(stage)
        eng.forward_windows(arena, start, valid, dst, keep, logits)
        ref = st[name].transpose(1, 2).contiguous() if stage <= 6 else st[name]
        got = torch.from_numpy(eng.debug_stage(stage, B, ref.numel()).reshape(ref.shape))
        err = (got - ref).abs().max().item()
        assert err <= (conv_tol if stage <= 6 else hid_tol), f"{name}: max abs err {err}"
    eng.debug_stop_after(-1)
    eng.forward_windows(arena, start, valid, dst, keep, logits)
    torch.cuda.synchronize()
    err = (logits.cpu() - st["logits"].reshape(B * T, 4)).abs().max().item()
    assert err <= (LOGIT_TOL if dtype == "f16" else 1.2e-1), err
    eng.close()


@pytest.mark.parametrize("gamma_scale", [1.0, 2000.0])
def test_conv0_quiet_loud_and_out_of_range_windows(weights, gamma_scale):
    """conv0 + GroupNorm + GELU (stage 0) on inputs that stress the fp16-split MFMA form: a -100 dBFS window, a loud
    window with a DC offset, and a near-silent window with one full-scale click.  With the GroupNorm weight scaled
    by 2000 the folded operand leaves the fp16 range for some windows, which must then take the VALU form."""
    chunk, B, W = 1.0, 3, 16000
    rng = np.random.default_rng(7)
    wav = np.zeros((B, W), dtype=np.float32)
    wav[0] = rng.standard_normal(W).astype(np.float32) * np.float32(1e-5)
    wav[1] = np.clip(rng.standard_normal(W).astype(np.float32) * np.float32(0.3) + np.float32(0.4), -1, 1)
    wav[2] = rng.standard_normal(W).astype(np.float32) * np.float32(1e-4)
    wav[2, 5000] = 1.0
    w = dict(weights)
    key = "feature_extractor.conv_layers.0.layer_norm.weight"
    w[key] = (np.asarray(w[key]) * np.float32(gamma_scale)).astype(np.float32) if not isinstance(w[key], torch.Tensor) \
        else w[key] * gamma_scale
    with torch.no_grad():
        ref = O.encoder_stages(w, torch.from_numpy(wav))["conv0"].transpose(1, 2).contiguous()
    eng = _engine(w, chunk, B)
    T = O.conv_out_len(W)
    logits = torch.zeros((B * T, 4), device="cuda")
    eng.debug_stop_after(0)
    eng.forward_windows(torch.from_numpy(wav.reshape(-1).copy()).cuda(), np.arange(B) * W, np.full(B, W), np.arange(B) * T,
                        np.full(B, T), logits)
    got = torch.from_numpy(eng.debug_stage(0, B, ref.numel()).reshape(ref.shape))
    ok = ref.abs() < 6.0e4                       # beyond that the 16-bit activation itself overflows
    assert ok.float().mean().item() > 0.99
    err = ((got - ref).abs() - 1.5e-3 * ref.abs())[ok].max().item()     # fp16 storage: 2^-11 relative, plus 2e-3 abs
    assert err <= 2e-3, err
    eng.close()


def test_golden_logits_fixture(weights, golden_dir):
    """Committed fixture (tests/golden/encoder_1s.npz), no oracle code involved in the expected values."""
    info = json.loads((golden_dir / "golden_encoder.json").read_text())
    z = np.load(golden_dir / "encoder_1s.npz")
    rng = np.random.default_rng(info["seed_input"])
    wav = rng.standard_normal((2, 16000), dtype=np.float32) * np.float32(0.1)
    wav[1, 9000:] = 0.0
    eng = _engine(O.make_weights(seed=info["seed_weights"]), 1.0, 2)
    T = eng.frames_per_window
    logits = torch.zeros((2 * T, 4), device="cuda")
    eng.forward_windows(torch.from_numpy(wav.reshape(-1).copy()).cuda(), [0, 16000], [16000, 16000], [0, T], [T, T], logits)
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy().reshape(2, T, 4), z["logits"], atol=LOGIT_TOL, rtol=0)
    eng.close()


# ----------------------------------------------------------------------------- whole files
def _masks_and_segments(logits_cpu, uid="f"):
    mask = O.apply_thresholds(logits_cpu, THR)
    return mask, O.intervals_to_segments(O.create_intervals(mask), uid)


@pytest.mark.parametrize("chunk,seconds", [(4.0, 30.7), (10.0, 47.3), (15.0, 36.2)])
def test_file_logits_labels_and_segments(weights, chunk, seconds):
    """Multi-window file with a ragged tail: tiling, stitching, logits, labels, segments."""
    wav = synth.synth_audio(seconds, seed=11)
    ref = O.apply_model_on_waveform(weights, wav, chunk, batch_size=4)
    eng = _engine(weights, chunk, 3)        # batch 3 -> several forward calls + a short last batch
    got = eng.infer_waveform(torch.from_numpy(wav).cuda(), batch_size=3)
    torch.cuda.synchronize()
    assert got.shape == ref.shape == (O.n_frames_for_samples(len(wav)), 4)
    diff = (got.cpu() - ref).abs()
    assert diff.max().item() <= LOGIT_TOL, diff.max().item()
    m_ref, seg_ref = _masks_and_segments(ref)
    m_got, seg_got = _masks_and_segments(got.cpu())
    agree = (m_ref == m_got).float().mean().item()
    assert agree >= AGREEMENT, agree
    confident = ref.abs() > LOGIT_TOL
    assert torch.equal(m_ref[confident], m_got[confident])
    if torch.equal(m_ref, m_got):
        assert seg_ref == seg_got
    # GPU decode of the GPU logits == oracle decode of the same logits (index work: bit exact)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    s, e, lab, max_logit, mean_sig = eng.decode(got.contiguous(), lo, hi)
    have = [(int(a), int(b), O.DEFAULT_LABELS[int(c)]) for a, b, c in zip(s, e, lab)]
    assert have == O.create_intervals(m_got)
    mx, mean = O.sigmoid_summary(got.cpu())
    assert round(float(torch.sigmoid(torch.tensor(max_logit)).item()), 4) == mx
    assert abs(round(mean_sig, 4) - mean) <= 1e-4
    eng.close()


def test_short_and_edge_lengths(weights):
    eng = _engine(weights, 1.0, 4)
    with pytest.raises(lib.VtcError) as ei:
        eng.infer_waveform(torch.zeros(399, device="cuda"))
    assert ei.value.status == lib.ERR_SHORT and "too short" in str(ei.value)
    T = eng.frames_per_window
    for n in (400, 719, 720, 16000, 16001, T * 320 + 399, T * 320 + 400, 3 * T * 320 + 777):
        wav = (np.random.default_rng(n).standard_normal(n) * 0.1).astype(np.float32)
        got = eng.infer_waveform(torch.from_numpy(wav).cuda())
        ref = O.apply_model_on_waveform(weights, wav, 1.0, batch_size=4)
        assert got.shape == ref.shape, n
        assert (got.cpu() - ref).abs().max().item() <= LOGIT_TOL, n
    with pytest.raises(ValueError):
        eng.infer_waveform(torch.zeros(1000, device="cuda", dtype=torch.float64))
    with pytest.raises(lib.VtcError):
        eng.forward_windows(torch.zeros(16000, device="cuda"), [0] * 5, [16000] * 5, [0] * 5, [T] * 5,
                            torch.zeros((5 * T, 4), device="cuda"))   # more windows than max_batch
    eng.close()


# ----------------------------------------------------------------------------- full-size properties
def test_full_size_batch64_10s_properties(weights):
    """BASELINE configs[1] shape: 64 windows x 10 s.  Determinism, batch-position invariance (bit exact),
    stitching layout, and two windows checked against the oracle."""
    chunk, B = 10.0, 64
    eng = _engine(weights, chunk, B)
    T, W, step = eng.frames_per_window, eng.window_samples, eng.step
    assert (T, W, step) == (499, 160000, 159680)
    n = (B - 1) * step + W - 12345                      # ragged tail in the last window
    wav = synth.synth_audio(n / 16000.0, seed=21)[:n]
    gpu = torch.from_numpy(wav).cuda()
    a = eng.infer_waveform(gpu)
    b = eng.infer_waveform(gpu)
    torch.cuda.synchronize()
    assert a.shape[0] == O.n_frames_for_samples(n)
    assert torch.equal(a, b), "two runs differ: the path must be deterministic (reference tests/test_reproducibility.py:212)"
    assert torch.isfinite(a).all()
    # the same window alone (batch of 1, other position) gives the same bits
    for wi in (0, 17, 62):
        solo = torch.zeros((T, 4), device="cuda")
        eng.forward_windows(gpu, [wi * step], [W], [0], [T], solo)
        torch.cuda.synchronize()
        assert torch.equal(solo, a[wi * T:(wi + 1) * T]), f"window {wi} depends on its batch position"
    # oracle on two windows (first, and the zero-padded tail)
    plan = O.plan_windows(n, chunk)
    for wi in (0, B - 1):
        x = np.zeros(W, np.float32)
        x[: plan.valid[wi]] = wav[plan.starts[wi]: plan.starts[wi] + plan.valid[wi]]
        ref = O.encoder_logits(weights, torch.from_numpy(x[None]))[0][: plan.keep[wi]]
        got = a[wi * T: wi * T + plan.keep[wi]].cpu()
        assert (got - ref).abs().max().item() <= LOGIT_TOL
        assert ((got > 0) == (ref > 0)).float().mean().item() >= AGREEMENT
    eng.close()


def test_daylong_16h_recording_properties(weights):
    """BASELINE configs[3]: one 16 h recording (921.6 M samples, 5,772 windows of 10 s, 2.88 M frames) on one GPU.
    The oracle cannot run this size, so the checks are size-independent properties: the signal repeats with the window
    step, hence every full window must reproduce window 0 bit for bit wherever it sits in a batch or in the file
    (tiling + stitching), window 0 itself is checked against the oracle, and the GPU decode of all 2.88 M frames must
    equal a numpy run-length decode of the same logits."""
    chunk, B = 10.0, 64
    eng = _engine(weights, chunk, B)
    T, W, step = eng.frames_per_window, eng.window_samples, eng.step
    n = 16 * 3600 * 16000
    period = torch.from_numpy(synth.synth_audio(step / 16000.0, seed=16)[:step]).cuda()
    assert period.numel() == step
    reps = -(-n // step)
    wav = period.repeat(reps)[:n].contiguous()
    logits = eng.infer_waveform(wav, batch_size=B)
    torch.cuda.synchronize()
    assert logits.shape == (O.n_frames_for_samples(n), 4) == (2_879_999, 4)
    full = (n - W) // step + 1                                   # windows that lie entirely inside the recording
    assert full == 5771
    tiled = logits[: full * T].view(full, T, 4)
    assert torch.equal(tiled, tiled[0:1].expand(full, T, 4))
    ref0 = O.apply_model_on_waveform(weights, wav[:W].cpu().numpy(), chunk)[:T]
    assert (tiled[0].cpu() - ref0).abs().max().item() <= LOGIT_TOL
    # the ragged tail window (zero-padded, so its statistics differ from a full window): against the oracle
    tail = logits[full * T:]
    ref_tail = O.apply_model_on_waveform(weights, wav[full * step:].cpu().numpy(), chunk)
    assert tail.shape == ref_tail.shape == (O.n_frames_for_samples(n - full * step), 4)
    assert (tail.cpu() - ref_tail).abs().max().item() <= LOGIT_TOL
    # decode of the whole recording vs numpy on the same logits
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    s, e, lab, max_logit, mean_sig = eng.decode(logits.contiguous(), lo, hi)
    host = logits.cpu().numpy()
    at = 0
    for l in range(4):
        act = (host[:, l] >= lo[l]) & (host[:, l] <= hi[l])
        d = np.diff(np.concatenate([[0], act.astype(np.int8), [0]]))
        st, en = np.flatnonzero(d == 1), np.flatnonzero(d == -1) - 1
        k = len(st)
        assert np.array_equal(s[at:at + k], st * 320) and np.array_equal(e[at:at + k], en * 320 + 400)
        assert np.all(lab[at:at + k] == l)
        at += k
    assert at == len(s)
    assert max_logit == float(host.max())
    assert abs(mean_sig - float((1.0 / (1.0 + np.exp(-host.astype(np.float64)))).mean())) <= 1e-6
    eng.close()


def test_large_batch_outputs_over_2gib(weights):
    """Batch 400 x 4 s windows: the conv1 / conv2 outputs exceed 2 GiB, which the persistent GEMM's 32-bit buffer
    offsets cannot address in one launch; gemm_go cuts such a GEMM into row chunks of whole windows.  Periodic input,
    so every window (on either side of a chunk boundary) must reproduce window 0 bit for bit."""
    chunk, B = 4.0, 400
    eng = _engine(weights, chunk, B)
    T, W, step = eng.frames_per_window, eng.window_samples, eng.step
    assert (T, W, step) == (199, 64000, 63680)
    period = torch.from_numpy(synth.synth_audio(step / 16000.0, seed=23)[:step]).cuda()
    wav = period.repeat(B + 1)[: (B - 1) * step + W].contiguous()
    logits = eng.infer_waveform(wav, batch_size=B)
    torch.cuda.synchronize()
    assert logits.shape == (B * T, 4)
    tiled = logits.view(B, T, 4)
    assert torch.equal(tiled, tiled[0:1].expand(B, T, 4))
    ref0 = O.apply_model_on_waveform(weights, wav[:W].cpu().numpy(), chunk)
    assert (tiled[0].cpu() - ref0).abs().max().item() <= LOGIT_TOL
    eng.close()


def test_row_chunked_layernorm_gemms_equal_the_unchunked_ones(weights, monkeypatch):
    """The LayerNorm-input / LayerNorm-residual GEMMs cut outputs over 2 GiB into row chunks and move every per-row array
    (row statistics, residual rows, partial statistics) with the chunk; that only triggers above ~349 k rows.  The test hook
    VTC_GEMM_ROWS_MAX forces chunks of 512 / 1280 rows on a small batch: logits must equal the unchunked run bit for bit
    (M = 24 x 49 = 1176 rows: 3 chunks with a ragged last one, and 1 + a ragged one)."""
    eng = _engine(weights, 1.0, 24)
    wav = torch.from_numpy(synth.synth_audio(24.5, seed=31)).cuda()
    monkeypatch.delenv("VTC_GEMM_ROWS_MAX", raising=False)
    whole = eng.infer_waveform(wav, batch_size=24).clone()
    torch.cuda.synchronize()
    for rows in ("512", "1280", "256"):
        monkeypatch.setenv("VTC_GEMM_ROWS_MAX", rows)
        cut = eng.infer_waveform(wav, batch_size=24)
        torch.cuda.synchronize()
        assert torch.equal(cut, whole), rows
    monkeypatch.delenv("VTC_GEMM_ROWS_MAX")
    eng.close()


def test_bf16_mode_error_is_measured_and_bounded(weights):
    """bf16 MFMA operands cannot meet the 2e-2 bar on this model (DESIGN.md 'Precision'); the mode stays
    selectable and its error is pinned here so a regression shows.  The agreement bar is 0.97 on this 12 s clip of
    stress-family weights (2,400 frame-labels, one label is 0.0004): round 2 moved it from 0.98 when the layers became
    LayerNorm-free, and round 3 measured that step on the C1-size recordings with both libraries (profiles/
    r03_bf16_before_after.json, 120 k frame-labels per case): mean |dlogit| 0.0153-0.0158 before, 0.0155-0.0160 after
    (+1.4 %), agreement 0.99702-0.99824 before, 0.99692-0.99812 after, max error 0.086-0.118 before, 0.084-0.096 after.
    bf16's error is set by its 8-bit operand mantissa in the conv and FFN contractions, not by where the residual stream is
    rounded; on a clip this short the 0.98 / 0.97 difference is a handful of labels."""
    wav = synth.synth_audio(12.0, seed=5)
    ref = O.apply_model_on_waveform(weights, wav, 4.0, batch_size=4)
    eng = _engine(weights, 4.0, 4, "bf16")
    got = eng.infer_waveform(torch.from_numpy(wav).cuda()).cpu()
    err = (got - ref).abs().max().item()
    assert err <= 1.5e-1, err
    assert ((got > 0) == (ref > 0)).float().mean().item() >= 0.97
    eng.close()


# ----------------------------------------------------------------------------- decode
def _decode_case(eng, logits_np, thr):
    lo, hi = logit_bounds(thr, O.DEFAULT_LABELS)
    t = torch.from_numpy(logits_np).cuda()
    s, e, lab, max_logit, mean_sig = eng.decode(t, lo, hi)
    have = [(int(a), int(b), O.DEFAULT_LABELS[int(c)]) for a, b, c in zip(s, e, lab)]
    want = O.create_intervals(O.apply_thresholds(torch.from_numpy(logits_np), thr))
    assert have == want
    mx, mean = O.sigmoid_summary(torch.from_numpy(logits_np))
    assert round(float(torch.sigmoid(torch.tensor(max_logit)).item()), 4) == mx
    assert abs(round(mean_sig, 4) - mean) <= 1e-4
    return have


def test_decode_bit_exact_against_oracle(weights):
    eng = _engine(weights, 1.0, 1)
    rng = np.random.default_rng(3)
    grid = lambda a: (np.round(a * 1024) / 1024).astype(np.float32)   # keeps logits off the 1-ulp sigmoid boundary
    mixed = dict(THR)
    mixed["MAL"] = {"lower_bound": 0.3, "upper_bound": 0.9}
    mixed["OCH"] = {"lower_bound": 0.0, "upper_bound": 1.0}
    cases = [
        grid(rng.standard_normal((1, 4))), grid(rng.standard_normal((255, 4))), grid(rng.standard_normal((256, 4))),
        grid(rng.standard_normal((257, 4))), grid(rng.standard_normal((100_003, 4)) * 2),
        np.full((1000, 4), 3.0, np.float32), np.full((1000, 4), -3.0, np.float32),
        np.tile(np.array([[1.0, -1.0, 1.0, -1.0], [-1.0, 1.0, 1.0, -1.0]], np.float32), (700, 1)),
        np.zeros((513, 4), np.float32),                        # sigmoid(0) == 0.5 -> active (inclusive bound)
    ]
    smooth = np.cumsum(rng.standard_normal((300_000, 4)), axis=0) * 0.05   # long runs crossing many blocks
    cases.append(grid(smooth - smooth.mean(0)))
    for c in cases:
        for thr in (THR, mixed):
            _decode_case(eng, np.ascontiguousarray(c), thr)
    runs = _decode_case(eng, np.full((1000, 4), 3.0, np.float32), THR)
    assert runs == [(0, 999 * 320 + 400, l) for l in O.DEFAULT_LABELS]
    eng.close()


def test_many_one_frame_recordings_all_active(weights):
    """Runs are split at recording boundaries, so n frames in f recordings can hold up to (n + f) / 2 runs per label:
    2000 one-frame recordings, every frame active, give 2000 runs per label from 2000 frames (more than n / 2 + 1)."""
    eng = _engine(weights, 1.0, 1)
    n = 2000
    logits = torch.full((n, 4), 2.0, device="cuda")
    rows = np.arange(n + 1, dtype=np.int64)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    want_file = np.repeat(np.arange(n), 4)
    want_lab = np.tile(np.arange(4), n)
    for call in ("files", "launch"):
        if call == "files":
            fidx, s, e, lab, mx, mean = eng.decode_files(logits, rows, lo, hi, summary=True)
        else:
            eng.decode_launch(logits, rows, lo, hi, summary=True)
            fidx, s, e, lab, mx, mean = eng.decode_collect()
        assert len(s) == 4 * n
        assert np.array_equal(fidx, want_file) and np.array_equal(lab, want_lab)
        assert np.all(s == 0) and np.all(e == 400)
        assert np.all(mx == 2.0) and np.allclose(mean, 1 / (1 + math.exp(-2.0)))
    # mixed: recordings of 1..3 frames with alternating activity, against the per-recording oracle
    rng = np.random.default_rng(8)
    lens = rng.integers(1, 4, size=700)
    rows = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    host = np.where(rng.random((int(rows[-1]), 4)) < 0.6, 1.5, -1.5).astype(np.float32)
    fidx, s, e, lab = eng.decode_files(torch.from_numpy(host).cuda(), rows, lo, hi)
    have = [(int(f), int(a), int(b), O.DEFAULT_LABELS[int(c)]) for f, a, b, c in zip(fidx, s, e, lab)]
    want = []
    for f in range(len(lens)):
        m = O.apply_thresholds(torch.from_numpy(host[rows[f]:rows[f + 1]]), THR)
        want += [(f, a, b, l) for a, b, l in O.create_intervals(m)]
    assert have == want
    eng.close()


def test_non_finite_logits_are_an_error_not_silence(weights):
    """fp16 storage overflows at 65504; an inf there reaches the logits as NaN, and NaN compares false against every
    bound: without a guard the recording would decode as 'no speech'.  The decode must fail (single recording) or
    mark exactly the affected recordings (several), and the host turns that into an error row."""
    from vtc_amd.pipeline import vtc as cli
    eng = _engine(weights, 1.0, 2)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    host = np.random.default_rng(4).standard_normal((600, 4)).astype(np.float32)
    host[250, 2] = np.nan
    host[460, 0] = np.inf
    dev = torch.from_numpy(host).cuda()
    with pytest.raises(lib.VtcError) as ei:
        eng.decode(dev, lo, hi)
    assert ei.value.status == lib.ERR_NONFINITE and "non-finite" in str(ei.value)
    rows = np.array([0, 200, 300, 450, 600], dtype=np.int64)     # recordings 1 and 3 are affected
    for call in ("files", "launch"):
        if call == "files":
            fidx, s, e, lab, mx, mean = eng.decode_files(dev, rows, lo, hi, summary=True)
        else:
            eng.decode_launch(dev, rows, lo, hi, summary=True)
            fidx, s, e, lab, mx, mean = eng.decode_collect()
        assert [bool(np.isnan(v)) for v in mx] == [False, True, False, True]
        for f in (0, 2):     # untouched recordings: same runs as decoded alone
            a, b, c, _, _ = eng.decode(dev[rows[f]:rows[f + 1]].contiguous(), lo, hi)
            sel = fidx == f
            assert np.array_equal(s[sel], a) and np.array_equal(e[sel], b) and np.array_equal(lab[sel], c)
    with pytest.raises(lib.VtcError):
        eng.decode_files(dev, rows, lo, hi)          # no summary asked for: nothing could carry the mark
    eng.decode_launch(dev, rows, lo, hi)
    with pytest.raises(lib.VtcError):
        eng.decode_collect()
    eng.decode_launch(dev[:200].contiguous(), np.array([0, 200]), lo, hi)     # the engine is usable again afterwards
    assert len(eng.decode_collect()[1]) > 0
    eng.close()
    # a weight set that drives the FFN hidden state past 65504: the forward itself yields NaN logits
    w = dict(weights)
    key = "encoder.layers.5.feed_forward.intermediate_dense.weight"
    w[key] = w[key] * 1.0e5
    big = _engine(w, 1.0, 2)
    wav = synth.synth_audio(2.5, seed=3)
    got = big.infer_waveform(torch.from_numpy(wav).cuda())
    assert not torch.isfinite(got).all()
    with pytest.raises(ValueError, match="non-finite"):
        cli.process_file(big, wav, "loud", 0.5, list(O.DEFAULT_LABELS), 2, None)
    big.close()


@pytest.mark.parametrize("n_labels", [1, 3, 8])
def test_other_label_counts(n_labels):
    """The head width is data (cfg.data.classes, SURVEY 8c: label set unpinned): 1, 3 and the maximum of 8 labels
    through forward + decode against the oracle."""
    from vtc_amd.engine import VtcEngine
    labels = tuple(f"L{i}" for i in range(n_labels))
    w = O.make_weights(seed=5, n_labels=n_labels)
    wav = synth.synth_audio(5.3, seed=61)
    ref = O.apply_model_on_waveform(w, wav, 2.0)
    eng = VtcEngine(w, n_labels=n_labels, chunk_duration_s=2.0, max_batch=2, device=0)
    got = eng.infer_waveform(torch.from_numpy(wav).cuda())
    assert got.shape == ref.shape == (O.n_frames_for_samples(len(wav)), n_labels)
    assert (got.cpu() - ref).abs().max().item() <= LOGIT_TOL
    thr = {l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in labels}
    lo, hi = logit_bounds(thr, labels)
    s, e, lab, _, _ = eng.decode(got.contiguous(), lo, hi)
    mask = O.apply_thresholds(got.cpu(), thr, labels)
    assert [(int(a), int(b), labels[int(c)]) for a, b, c in zip(s, e, lab)] == O.create_intervals(mask, labels)
    eng.close()
    with pytest.raises(Exception):
        VtcEngine(w, n_labels=n_labels + 1, chunk_duration_s=2.0, max_batch=2, device=0)   # head shape mismatch


# ----------------------------------------------------------------------------- drop-in surface + CLI
def test_reference_host_flow_through_segma_surface(weights, tmp_path):
    """The reference's own call sequence (src/pipeline/vtc.py:183-195, 236-243, 84-96, 284-288)."""
    from vtc_amd.audio import write_wav
    from vtc_amd.core.intervals import intervals_to_segments
    from vtc_amd.core.metadata import vtc_meta_row
    from vtc_amd.segma_compat import install_as_segma
    install_as_segma()
    from segma.config import load_config
    from segma.inference import apply_model_on_audio, apply_thresholds, create_intervals
    from segma.models import Models
    from segma.utils.encoders import MultiLabelEncoder

    (tmp_path / "config.yml").write_text("model:\n  name: surgical_hydra\ndata:\n  classes: [FEM, MAL, KCHI, OCH]\n"
                                         "audio:\n  chunk_duration_s: 4.0\n")
    wav = synth.synth_audio(9.3, seed=2)
    write_wav(tmp_path / "a.wav", wav)
    cfg = load_config(tmp_path / "config.yml")
    enc = MultiLabelEncoder(labels=cfg.data.classes)
    model = Models[cfg.model.name].load_from_checkpoint(checkpoint_path="synthetic:0", label_encoder=enc, config=cfg, train=False)
    model.eval()
    model.to(torch.device("cuda"))
    with torch.no_grad():
        logits_t = apply_model_on_audio(audio_path=tmp_path / "a.wav", model=model, conv_settings=model.conv_settings,
                                        device="cuda", batch_size=128, chunk_duration_s=cfg.audio.chunk_duration_s)
    again = apply_model_on_audio(audio_path=tmp_path / "a.wav", model=model, conv_settings=model.conv_settings,
                                 device="cuda", batch_size=128, chunk_duration_s=cfg.audio.chunk_duration_s)
    assert torch.equal(logits_t, again)
    cpu = logits_t.cpu()
    thr = {l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in enc.labels}
    thresholded = apply_thresholds(cpu, thr, "cpu").detach()
    # a real 0/1 tensor the caller can look at ("apply_thresholds(logits, dict, 'cpu') -> Tensor")
    assert isinstance(thresholded, torch.Tensor) and thresholded.dtype == torch.uint8 and thresholded.device.type == "cpu"
    assert torch.equal(thresholded, O.apply_thresholds(cpu, THR))
    assert thresholded.float().mean().item() == O.apply_thresholds(cpu, THR).float().mean().item()
    intervals = create_intervals(thresholded, model.conv_settings, enc)
    assert intervals == O.create_intervals(O.apply_thresholds(cpu, THR))
    # plain tensors made by the caller work too: logits without provenance, and a mask changed after the fact
    plain = cpu.clone().as_subclass(torch.Tensor)
    assert torch.equal(apply_thresholds(plain, thr, "cpu"), thresholded)
    edited = thresholded.clone()
    edited[:, 1] = 0
    edited[5:9, 2] = 1
    assert create_intervals(edited, model.conv_settings, enc) == O.create_intervals(edited.numpy())
    in_place = apply_thresholds(cpu, thr, "cpu")
    in_place[:, 0] = 0                       # the carried device logits no longer describe this mask
    assert create_intervals(in_place, model.conv_settings, enc) == O.create_intervals(in_place.numpy())
    # two models alive at once: each mask decodes on the engine that made its logits
    enc2 = MultiLabelEncoder(labels=["A", "B"])
    model2 = Models["hydra"].load_from_checkpoint(checkpoint_path="synthetic:3", label_encoder=enc2, config=cfg, train=False)
    l2 = apply_model_on_audio(audio_path=tmp_path / "a.wav", model=model2.to(torch.device("cuda")),
                              conv_settings=model2.conv_settings, device="cuda", batch_size=8, chunk_duration_s=2.0)
    m2 = apply_thresholds(l2.cpu(), {l: {"lower_bound": 0.4, "upper_bound": 1.0} for l in enc2.labels}, "cpu")
    assert m2.shape == (l2.shape[0], 2)
    assert create_intervals(m2, model2.conv_settings, enc2) == O.create_intervals(m2.numpy(), labels=("A", "B"))
    assert create_intervals(thresholded, model.conv_settings, enc) == intervals
    from vtc_amd.audio import read_audio
    ref = O.apply_model_on_waveform(weights, read_audio(tmp_path / "a.wav"), 4.0)
    assert (cpu - ref).abs().max().item() <= LOGIT_TOL
    segs = intervals_to_segments(intervals, "a")
    row = vtc_meta_row("a", 0.5, segs, *O.sigmoid_summary(cpu))
    assert row["vtc_n_segments"] == len(segs)


def test_cli_outputs_resume_and_error_rows(weights, tmp_path, monkeypatch):
    from vtc_amd.audio import read_audio, write_wav
    from vtc_amd.pipeline import vtc as cli
    from vtc_amd import utils
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    durs = {"f0": 3.2, "f1": 7.9, "f2": 0.02, "f3": 12.0, "f4": 1.0}     # f2 is shorter than the receptive field
    for i, (uid, d) in enumerate(durs.items()):
        write_wav(tmp_path / "audio" / f"{uid}.wav", synth.synth_audio(d, seed=40 + i))
    (tmp_path / "manifests" / "ds.csv").write_text("path\n" + "\n".join(str(tmp_path / "audio" / f"{u}.wav") for u in durs) + "\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    common = dict(config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=4)
    cli.main("ds", array_id=0, array_count=2, save_logits=True, **common)
    cli.main("ds", array_id=1, array_count=2, **common)
    out = tmp_path / "output" / "ds"
    meta = {r["uid"]: r for f in sorted((out / "vtc_meta").glob("shard_*.parquet")) for r in pq.read_table(f).to_pylist()}
    assert set(meta) == set(durs)
    assert "too short" in meta["f2"]["error"] and math.isnan(meta["f2"]["vtc_threshold"]) and meta["f2"]["vtc_n_segments"] == 0
    raw = [r for f in sorted((out / "vtc_raw").glob("shard_*.parquet")) for r in pq.read_table(f).to_pylist()]
    merged_t = [pq.read_table(f) for f in sorted((out / "vtc_merged").glob("shard_*.parquet"))]
    assert pq.read_table(out / "vtc_raw" / "shard_0.parquet").schema.names == ["uid", "onset", "offset", "duration", "label"]
    assert pq.read_table(out / "vtc_meta" / "shard_0.parquet").schema.names == list(utils.META_SCHEMA.names)
    # against the oracle run on the same files (confident frames decide; identical when masks are identical)
    for uid in ("f0", "f1", "f3", "f4"):
        wav = read_audio(tmp_path / "audio" / f"{uid}.wav")
        ref = O.apply_model_on_waveform(weights, wav, 4.0)
        m = O.apply_thresholds(ref, THR)
        segs = O.intervals_to_segments(O.create_intervals(m), uid)
        mine = sorted([r for r in raw if r["uid"] == uid], key=lambda r: (r["label"], r["onset"]))
        assert meta[uid]["error"] == "" and meta[uid]["vtc_n_segments"] == len(mine)
        if (ref.abs() > LOGIT_TOL).all():
            assert mine == sorted(segs, key=lambda r: (r["label"], r["onset"]))
            want = O.vtc_meta_row(uid, 0.5, segs, *O.sigmoid_summary(ref))
            for k, v in want.items():
                if k in ("vtc_max_sigmoid", "vtc_mean_sigmoid"):      # sigmoid' <= 1/4 -> 2e-2 on logits is 5e-3 here
                    assert abs(meta[uid][k] - v) <= LOGIT_TOL / 4 + 1e-4, (uid, k)
                else:
                    assert meta[uid][k] == v, (uid, k)
    for t in merged_t:
        want = utils.merge_segments(pq.read_table(out / "vtc_raw" / "shard_0.parquet"), 0.3, 0.1) if t is merged_t[0] else None
        if want is not None:
            assert t.to_pylist() == want.to_pylist()
    saved = torch.load(out / "logits" / "f0-logits_dict_t.pt", weights_only=True)
    assert list(saved) == list(O.DEFAULT_LABELS) and saved["FEM"].shape[0] == O.n_frames_for_samples(int(3.2 * 16000))
    # resume: nothing is recomputed, outputs stay identical
    before = (out / "vtc_raw" / "shard_0.parquet").read_bytes()
    cli.main("ds", array_id=0, array_count=2, **common)
    assert pq.read_table(out / "vtc_raw" / "shard_0.parquet").to_pylist() == pq.read_table(pa_buf(before)).to_pylist()
    assert (tmp_path / "logs" / "benchmarks.jsonl").exists()


def test_cli_durable_progress_and_resume_after_crash(tmp_path, monkeypatch):
    """--flush_every: a run killed mid-shard keeps the flushed files, the rerun only does the rest, and the final parquet
    files equal those of an uninterrupted run (reference writes once at the end, vtc.py:375-387)."""
    from vtc_amd.audio import write_wav
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    uids = [f"g{i}" for i in range(6)]
    for i, u in enumerate(uids):
        write_wav(tmp_path / "audio" / f"{u}.wav", synth.synth_audio(1.5 + 0.7 * i, seed=70 + i))
    listing = "path\n" + "\n".join(str(tmp_path / "audio" / f"{u}.wav") for u in uids) + "\n"
    (tmp_path / "manifests" / "clean.csv").write_text(listing)
    (tmp_path / "manifests" / "crash.csv").write_text(listing)
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    common = dict(config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=4, flush_seconds=0.0)
    cli.main("clean", **common)

    real, calls = cli.finish_file, {"n": 0}

    def dying(*a, **k):
        calls["n"] += 1
        if calls["n"] == 5:
            raise KeyboardInterrupt("simulated kill")
        return real(*a, **k)

    monkeypatch.setattr(cli, "finish_file", dying)
    with pytest.raises(KeyboardInterrupt):
        cli.main("crash", flush_every=2, **common)
    monkeypatch.setattr(cli, "finish_file", real)
    out = tmp_path / "output" / "crash"
    flushed = pq.read_table(out / "vtc_meta" / "shard_0.parquet").column("uid").to_pylist()
    assert flushed == uids[:4]                      # two flushes of two files each survived the kill
    seen = []
    monkeypatch.setattr(cli, "stage_audio", lambda path, eng, _f=cli.stage_audio: (seen.append(path), _f(path, eng))[1])
    monkeypatch.setattr(cli, "stage_bundle", lambda paths, eng, _f=cli.stage_bundle: (seen.extend(paths), _f(paths, eng))[1])
    cli.main("crash", flush_every=2, **common)
    assert [str(p).rsplit("/", 1)[-1] for p in seen] == ["g4.wav", "g5.wav"]   # only the unfinished files were redone
    for kind in ("vtc_meta", "vtc_raw", "vtc_merged"):
        a = pq.read_table(tmp_path / "output" / "clean" / kind / "shard_0.parquet")
        b = pq.read_table(out / kind / "shard_0.parquet")
        assert a.schema.names == b.schema.names and a.to_pylist() == b.to_pylist(), kind


# ----------------------------------------------------------------------------- audio front-end
@pytest.mark.parametrize("sr,ch,dtype", [(16000, 1, "int16"), (16000, 2, "int16"), (44100, 2, "int16"), (48000, 1, "float32"),
                                         (8000, 1, "int16"), (22050, 3, "float32"), (32000, 1, "int16"),
                                         (16000, 1, "int24"), (48000, 2, "int24")])
def test_gpu_front_end_against_oracle(weights, sr, ch, dtype):
    """vtc_prepare_audio (channel mix + windowed-sinc resampling on the GPU) against the oracle's restatement of
    torchaudio.functional.resample and against the host route (audio.sinc_resample).  int16 -> float and the mono mix
    are exact; the filter is an fp32 sum of <= ~40 products, compared at 2e-6.  Parity with segma itself: unpinned."""
    from vtc_amd import audio
    rng = np.random.default_rng(sr + ch)
    n = int(1.37 * sr) + 13
    x = rng.standard_normal((n, ch)) * 0.2
    packed = None
    if dtype == "int24":       # packed 24-bit PCM as a WAV stores it (format 2): exact in float32, so the oracle takes the floats
        v = np.clip(np.round(x * 8388608), -8388608, 8388607).astype(np.int32)
        packed = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=-1).astype(np.uint8)
        pcm = (v.astype(np.float64) / 8388608.0).astype(np.float32)
        assert np.array_equal(audio.pcm24_to_float32(packed).reshape(n, ch), pcm)
    else:
        pcm = np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16) if dtype == "int16" else x.astype(np.float32)
    if ch == 1:
        pcm = pcm[:, 0].copy()
    mono = O.pcm_to_mono_float(pcm)
    ref = O.resample_sinc_hann(mono, sr)
    eng = _engine(weights, 1.0, 2)
    got = eng.prepare_audio(torch.from_numpy(packed if packed is not None else pcm).cuda(), sr).cpu().numpy()
    assert got.shape == ref.shape == (-(-16000 * n // sr),)
    if sr == 16000:
        assert np.array_equal(got, mono)                       # conversion and mix: bit exact
    else:
        assert np.abs(got - ref).max() <= 2e-6
        assert np.abs(got - audio.sinc_resample(mono, sr)).max() <= 2e-6
    eng.close()


def test_cli_24_bit_wav_files_cross_pcie_packed(weights, tmp_path, monkeypatch):
    """24-bit PCM WAVs (mono 16 kHz and stereo 48 kHz) through the CLI: read_pcm hands the packed bytes to the GPU front-end
    (vtc_prepare_audio format 2), no host conversion.  Rows must equal those of the same samples converted on the host
    (audio.read_audio for the mono file: the conversion is exact, so this is bit for bit)."""
    import struct
    from vtc_amd import audio
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()

    def wav24(path, v, sr):
        n, ch = v.shape
        b = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=-1).astype(np.uint8).tobytes()
        path.write_bytes(b"RIFF" + struct.pack("<I", 36 + len(b)) + b"WAVEfmt " +
                         struct.pack("<IHHIIHH", 16, 1, ch, sr, sr * ch * 3, ch * 3, 24) + b"data" + struct.pack("<I", len(b)) + b)

    x = synth.synth_audio(23.4, seed=77)
    mono = np.clip(np.round(x * 8388608.0), -8388608, 8388607).astype(np.int32)[:, None]
    wav24(tmp_path / "audio" / "m24.wav", mono, 16000)
    up = np.interp(np.arange(int(9.1 * 48000)) / 48000.0, np.arange(len(x)) / 16000.0, x)
    st = np.clip(np.round(np.stack([up, 0.5 * up], axis=1) * 8388608.0), -8388608, 8388607).astype(np.int32)
    wav24(tmp_path / "audio" / "s24.wav", st, 48000)
    import aifc
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with aifc.open(str(tmp_path / "audio" / "a24.aiff"), "wb") as f:      # the mono samples again, big-endian in an AIFF
            f.setnchannels(1); f.setsampwidth(3); f.setframerate(16000)
            f.writeframes(mono[:, 0].astype(">i4").view(np.uint8).reshape(-1, 4)[:, 1:].tobytes())
    (tmp_path / "manifests" / "w24.csv").write_text("path\n" + "".join(f"{tmp_path / 'audio' / n}\n" for n in ("m24.wav", "s24.wav", "a24.aiff")))
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    got, _ = audio.read_pcm(tmp_path / "audio" / "s24.wav")
    assert got.dtype == np.uint8 and got.shape == (len(st), 2, 3)
    cli.main("w24", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=8)
    out = tmp_path / "output" / "w24"
    seg = pq.read_table(out / "vtc_raw" / "shard_0.parquet").to_pylist()
    meta = pq.read_table(out / "vtc_meta" / "shard_0.parquet").to_pylist()
    assert [r["uid"] for r in meta] == ["m24", "s24", "a24"] and all(r["error"] == "" for r in meta)
    rows_of = lambda u: [(r["onset"], r["offset"], r["duration"], r["label"]) for r in seg if r["uid"] == u]
    assert rows_of("a24") == rows_of("m24") and len(rows_of("a24")) > 3
    eng = _engine(weights, 4.0, 8)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    for uid in ("m24", "s24"):
        if uid == "m24":
            wav = torch.from_numpy(audio.read_audio(tmp_path / "audio" / "m24.wav")).cuda()      # host route: exact conversion
        else:
            wav = eng.prepare_audio(torch.from_numpy(np.ascontiguousarray(got)).cuda(), 48000)
            host = audio.read_audio(tmp_path / "audio" / "s24.wav")                              # host mix + resample: fp32 sums
            assert np.abs(wav.cpu().numpy() - host).max() <= 2e-6
        s_, e_, l_, _, _ = eng.decode(eng.infer_waveform(wav, batch_size=8).contiguous(), lo, hi)
        want = O.intervals_to_segments([(int(a), int(b), O.DEFAULT_LABELS[int(c)]) for a, b, c in zip(s_, e_, l_)], uid)
        mine = [(r["onset"], r["offset"], r["duration"], r["label"]) for r in seg if r["uid"] == uid]
        assert mine == [(r["onset"], r["offset"], r["duration"], r["label"]) for r in want] and len(mine) > 3, uid
    eng.close()


def test_cli_accepts_stereo_44k_file(weights, tmp_path, monkeypatch):
    """A 44.1 kHz stereo WAV goes through the CLI (GPU front-end) and gives the logits of the oracle run on the
    oracle-resampled mono signal."""
    from scipy.io import wavfile
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    base = synth.synth_audio(6.0, seed=91)
    up = np.interp(np.arange(int(6.0 * 44100)) / 44100.0, np.arange(len(base)) / 16000.0, base).astype(np.float32)
    stereo = np.stack([up, 0.5 * up[::-1]], axis=1)
    pcm = np.clip(np.round(stereo * 32768), -32768, 32767).astype(np.int16)
    wavfile.write(tmp_path / "audio" / "s44.wav", 44100, pcm)
    (tmp_path / "manifests" / "hi.csv").write_text(f"path\n{tmp_path / 'audio' / 's44.wav'}\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    cli.main("hi", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=4, save_logits=True)
    saved = torch.load(tmp_path / "output" / "hi" / "logits" / "s44-logits_dict_t.pt", weights_only=True)
    got = torch.stack([saved[l] for l in O.DEFAULT_LABELS], dim=1)
    ref = O.apply_model_on_waveform(weights, O.resample_sinc_hann(O.pcm_to_mono_float(pcm), 44100), 4.0)
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= LOGIT_TOL
    meta = pq.read_table(tmp_path / "output" / "hi" / "vtc_meta" / "shard_0.parquet").to_pylist()
    assert meta[0]["uid"] == "s44" and meta[0]["error"] == ""


def test_cli_flac_and_aiff_inputs_equal_wav(weights, tmp_path, monkeypatch):
    """The same PCM16 samples as .wav, .flac (library decoder) and .aiff give identical segments and meta rows through
    the CLI; a 24-bit FLAC (float path) agrees with the oracle on its own samples; an .mp3 becomes an error row; an 8 kHz
    G.711 mu-law WAV gives the rows of its expansion stored as PCM16."""
    import aifc, warnings
    from scipy.io import wavfile
    from oracle import flac_codec as FL
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    x = synth.synth_audio(21.3, seed=77)
    up = np.interp(np.arange(int(21.3 * 22050)) / 22050.0, np.arange(len(x)) / 16000.0, x)
    pcm = np.clip(np.round(np.stack([up, 0.3 * up], axis=1) * 32768), -32768, 32767).astype(np.int16)
    wavfile.write(tmp_path / "audio" / "a_wav.wav", 22050, pcm)
    (tmp_path / "audio" / "a_flac.flac").write_bytes(FL.encode(pcm.astype(np.int64), 22050, subframe="lpc"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with aifc.open(str(tmp_path / "audio" / "a_aiff.aiff"), "wb") as f:
            f.setnchannels(2); f.setsampwidth(2); f.setframerate(22050)
            f.writeframes(pcm.astype(">i2").tobytes())
    x24 = np.clip(np.round(x.astype(np.float64) * (1 << 23)), -(1 << 23), (1 << 23) - 1).astype(np.int64)
    (tmp_path / "audio" / "b24.flac").write_bytes(FL.encode(x24, 16000, bps=24))
    (tmp_path / "audio" / "c.mp3").write_bytes(b"ID3" + bytes(64))
    # 8-bit G.711 mu-law at 8 kHz (telephone speech) and the same samples, expanded with the Recommendation's table, as PCM16
    import struct
    from vtc_amd.audio import g711_table
    x8 = np.interp(np.arange(int(21.3 * 8000)) / 8000.0, np.arange(len(x)) / 16000.0, x)
    lin = np.clip(np.round(x8 * 32768), -32635, 32635).astype(np.int32)
    mag = np.abs(lin) + 0x84
    e = np.maximum(np.floor(np.log2(mag)).astype(np.int32) - 7, 0)
    codes = (~(np.where(lin < 0, 0x80, 0) | (e << 4) | ((mag >> (e + 3)) & 0x0F)) & 0xFF).astype(np.uint8)
    fmt = struct.pack("<HHIIHHH", 7, 1, 8000, 8000, 1, 8, 0)
    body = b"WAVEfmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", codes.size) + codes.tobytes()
    (tmp_path / "audio" / "d_ulaw.wav").write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    wavfile.write(tmp_path / "audio" / "d_lin.wav", 8000, g711_table("ulaw")[codes])
    names = ["a_wav.wav", "a_flac.flac", "a_aiff.aiff", "b24.flac", "c.mp3", "d_ulaw.wav", "d_lin.wav"]
    (tmp_path / "manifests" / "mix.csv").write_text("path\n" + "".join(f"{tmp_path / 'audio' / n}\n" for n in names))
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    cli.main("mix", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=8, save_logits=True)
    out = tmp_path / "output" / "mix"
    seg = pq.read_table(out / "vtc_raw" / "shard_0.parquet").to_pylist()
    meta = {r["uid"]: r for r in pq.read_table(out / "vtc_meta" / "shard_0.parquet").to_pylist()}
    by = lambda uid: [(r["onset"], r["offset"], r["label"]) for r in seg if r["uid"] == uid]
    assert len(by("a_wav")) > 0 and by("a_wav") == by("a_flac") == by("a_aiff")
    for k in ("vtc_speech_dur", "vtc_n_segments", "vtc_label_counts", "vtc_max_sigmoid", "vtc_mean_sigmoid", "error"):
        assert meta["a_wav"][k] == meta["a_flac"][k] == meta["a_aiff"][k], k
    la, lb = (torch.load(out / "logits" / f"{u}-logits_dict_t.pt", weights_only=True) for u in ("a_wav", "a_flac"))
    assert all(torch.equal(la[l], lb[l]) for l in O.DEFAULT_LABELS)
    saved = torch.load(out / "logits" / "b24-logits_dict_t.pt", weights_only=True)
    got = torch.stack([saved[l] for l in O.DEFAULT_LABELS], dim=1)
    ref = O.apply_model_on_waveform(weights, (x24 / float(1 << 23)).astype(np.float32), 4.0)
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= LOGIT_TOL
    assert "codec" in meta["c"]["error"] and meta["b24"]["error"] == ""
    assert len(by("d_ulaw")) > 0 and by("d_ulaw") == by("d_lin") and meta["d_ulaw"]["error"] == ""
    la, lb = (torch.load(out / "logits" / f"{u}-logits_dict_t.pt", weights_only=True) for u in ("d_ulaw", "d_lin"))
    assert all(torch.equal(la[l], lb[l]) for l in O.DEFAULT_LABELS)


# ----------------------------------------------------------------------------- several recordings per call (clips)
def test_multi_recording_batch_equals_per_recording(weights):
    """vtc_infer_waveforms / vtc_decode_files: windows of many short recordings share forward passes; rows and decoded
    runs must equal the per-recording calls bit for bit (windows are independent), and runs must not cross recordings."""
    eng = _engine(weights, 4.0, 8)
    secs = [0.02, 1.0, 3.99, 4.0, 4.03, 9.7, 0.5, 13.2, 2.2]          # first one is shorter than the receptive field
    wavs = [torch.from_numpy(synth.synth_audio(d, seed=200 + i)).cuda() for i, d in enumerate(secs)]
    logits, rows = eng.infer_waveforms(wavs, batch_size=8)
    assert rows[0] == 0 and rows[1] == 0                               # too short: zero rows, no error
    assert [int(rows[i + 1] - rows[i]) for i in range(len(secs))] == [O.n_frames_for_samples(int(w.numel())) for w in wavs]
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    fidx, s, e, lab = eng.decode_files(logits.contiguous(), rows, lo, hi)
    assert list(fidx) == sorted(fidx)
    for i, w in enumerate(wavs):
        mine = logits[int(rows[i]):int(rows[i + 1])]
        if i == 0:
            assert mine.shape[0] == 0
            continue
        single = eng.infer_waveform(w, batch_size=8)
        assert torch.equal(mine, single), i
        ss, se, sl, _, _ = eng.decode(single.contiguous(), lo, hi)
        sel = fidx == i
        assert list(s[sel]) == list(ss) and list(e[sel]) == list(se) and list(lab[sel]) == list(sl), i
        ref = O.apply_model_on_waveform(weights, w.cpu().numpy(), 4.0)
        assert (mine.cpu() - ref).abs().max().item() <= LOGIT_TOL
    # two-step form (vtc_decode_launch / vtc_decode_collect): same runs and summaries, also with a forward enqueued between
    # the two steps; one decode in flight at a time
    want = eng.decode_files(logits.contiguous(), rows, lo, hi, summary=True)
    eng.decode_launch(logits.contiguous(), rows, lo, hi, summary=True)
    with pytest.raises(RuntimeError, match="not been collected"):
        eng.decode_launch(logits.contiguous(), rows, lo, hi)
    other = eng.infer_waveform(wavs[7], batch_size=8)                 # queued behind the decode, runs while we collect
    got = eng.decode_collect()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    with pytest.raises(RuntimeError, match="without vtc_decode_launch"):
        eng._pending_decode = (1, False, None)
        eng.decode_collect()
    single = eng.infer_waveform(wavs[5], batch_size=8).contiguous()
    ss, se, sl, smax, smean = eng.decode(single, lo, hi)
    eng.decode_launch(single, [0, single.shape[0]], lo, hi, summary=True)
    f1, s1, e1, l1, mx1, mean1 = eng.decode_collect()
    assert not f1.any() and list(s1) == list(ss) and list(e1) == list(se) and list(l1) == list(sl)
    assert float(mx1[0]) == smax and float(mean1[0]) == smean
    eng.decode_launch(single, [0, single.shape[0]], lo, hi)                    # error paths drop a decode in flight
    eng.decode_abandon()
    eng.decode_abandon()                                                       # nothing pending: no-op
    eng.decode_launch(logits.contiguous(), [0, 0, 0], lo, hi, summary=True)   # recordings without rows
    f0, s0, e0, l0, mx0, mean0 = eng.decode_collect()
    assert len(f0) == 0 and list(mean0) == [0.0, 0.0] and np.isneginf(mx0).all()
    assert torch.equal(other, eng.infer_waveform(wavs[7], batch_size=8))
    # a boundary inside an active run splits it: all-active logits, two recordings of 5 and 7 frames
    hot = torch.full((12, 4), 5.0, device="cuda")
    f2, s2, e2, l2 = eng.decode_files(hot, np.array([0, 5, 12]), lo, hi)
    assert list(f2) == [0] * 4 + [1] * 4 and list(s2) == [0] * 8
    assert list(e2) == [4 * 320 + 400] * 4 + [6 * 320 + 400] * 4 and list(l2) == [0, 1, 2, 3] * 2
    eng.close()


def test_clips_caller_on_tar_shards(weights, tmp_path, monkeypatch):
    """vtc_amd.analysis.vtc_on_clips (reference src/analysis/vtc_on_clips.py): WebDataset tars in, clip-relative
    segments out, resume by clip_id, per-clip errors counted and skipped."""
    import io, tarfile
    from scipy.io import wavfile
    from oracle import flac_codec as FL
    from vtc_amd.analysis import vtc_on_clips as clips
    monkeypatch.chdir(tmp_path)
    shard_dir = tmp_path / "output" / "ds" / "shards"
    shard_dir.mkdir(parents=True)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "manifests" / "ds.csv").write_text("path\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    audio = {}

    def add(tf, name, data):
        info = tarfile.TarInfo(name)
        info.size = len(data)
        tf.addfile(info, io.BytesIO(data))

    def wav_bytes(x, sr=16000):
        buf = io.BytesIO()
        wavfile.write(buf, sr, np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16))
        return buf.getvalue()

    k = 0
    for t in range(2):
        with tarfile.open(shard_dir / f"shards-{t:04d}.tar", "w") as tf:
            for c in range(5):
                cid = f"uidA_{k:04d}"
                x = synth.synth_audio(2.0 + 1.7 * c, seed=300 + k)
                audio[cid] = np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
                if c % 2 == 1:                                  # the reference's packer writes flac by default
                    add(tf, f"{cid}.flac", FL.encode(audio[cid].astype(np.int64), 16000, subframe="lpc" if c == 1 else "auto"))
                else:
                    add(tf, f"{cid}.wav", wav_bytes(x))
                if c == 3:                                      # both members: flac wins (vtc_on_clips.py:163-166)
                    add(tf, f"{cid}.wav", wav_bytes(np.zeros_like(x)))
                add(tf, f"{cid}.json", b"{}")
                k += 1
                if t == 1 and c == 2:                           # a clip that cannot join a staging bundle, in the middle of one
                    up = np.interp(np.arange(int(5.3 * 44100)) / 44100.0, np.arange(len(x)) / 16000.0, x).astype(np.float32)
                    buf = io.BytesIO()
                    wavfile.write(buf, 44100, np.clip(np.round(np.stack([up, 0.4 * up], axis=1) * 32768), -32768, 32767).astype(np.int16))
                    add(tf, "uidC_0000.wav", buf.getvalue())
            if t == 0:
                add(tf, "uidB_0000.flac", b"fLaC not decodable here")      # error, skipped
                add(tf, "uidB_0001.json", b"{}")                            # no audio, error
    common = dict(config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=8)
    clips.main("ds", **common)
    out = tmp_path / "output" / "ds" / "vtc_clips"
    raw = pq.read_table(out / "vtc_raw" / "shard_0.parquet")
    assert raw.schema.names == ["clip_id", "onset", "offset", "duration", "label"]
    rows = raw.to_pylist()
    assert {r["clip_id"] for r in rows} <= set(audio) | {"uidC_0000"}
    for cid, pcm in audio.items():
        ref = O.apply_model_on_waveform(weights, O.pcm_to_mono_float(pcm), 4.0)
        if not (ref.abs() > LOGIT_TOL).all():
            continue
        want = O.intervals_to_segments(O.create_intervals(O.apply_thresholds(ref, THR)), cid)
        mine = [dict(uid=r["clip_id"], onset=r["onset"], offset=r["offset"], duration=r["duration"], label=r["label"])
                for r in rows if r["clip_id"] == cid]
        key = lambda r: (r["label"], r["onset"])
        assert sorted(mine, key=key) == sorted(want, key=key), cid
    merged = pq.read_table(out / "vtc_merged" / "shard_0.parquet")
    assert merged.schema.names == raw.schema.names and merged.num_rows <= raw.num_rows
    # one clip per forward, staged one by one (the reference's pattern): the same tables as packed forwards over bundled staging
    import shutil
    shutil.copytree(shard_dir, tmp_path / "output" / "ds1" / "shards")
    (tmp_path / "manifests" / "ds1.csv").write_text("path\n")
    clips.main("ds1", per_clip=True, **common)
    for kind in ("vtc_raw", "vtc_merged"):
        assert pq.read_table(tmp_path / "output" / "ds1" / "vtc_clips" / kind / "shard_0.parquet").to_pylist() == \
            pq.read_table(out / kind / "shard_0.parquet").to_pylist(), kind
    # resume: nothing with segments is recomputed
    calls = []
    monkeypatch.setattr(clips, "pack_forward", lambda eng, carry, nw, _f=clips.pack_forward: (calls.append([o.i for o in carry]), _f(eng, carry, nw))[1])
    clips.main("ds", **common)
    done = {r["clip_id"] for r in rows}
    assert all(c not in done for g in calls for c in g)
    assert pq.read_table(out / "vtc_raw" / "shard_0.parquet").to_pylist() == rows


def test_grid_cut_clips_reproduce_full_file_rows_and_align(weights, tmp_path, monkeypatch):
    """Clip-grid rule (src/packaging/clips.py:48-61) + the alignment consumer (src/pipeline/vtc_clip_alignment.py): a clip
    cut at a multiple of the window step re-creates the full-file windows, so its logits ARE the full-file rows (bit for
    bit); run through both CLIs (full files; FLAC clips in tars) the raw segments cover exactly the same time."""
    import io, tarfile
    import pyarrow as pa
    from oracle import flac_codec as FL
    from vtc_amd.analysis import vtc_on_clips as clips
    from vtc_amd.packaging.clips import CHUNK_STEP_F, snap_to_grid
    from vtc_amd.pipeline import vtc as cli, vtc_clip_alignment as A
    monkeypatch.chdir(tmp_path)
    step, T = CHUNK_STEP_F, 199
    n = 15 * step + 320
    dur = n / 16000.0
    cuts = [0.0] + [snap_to_grid(t, dur) for t in (20.0, 41.5)] + [dur]
    assert [round(c * 16000) for c in cuts[1:3]] == [5 * step, 10 * step]
    eng = _engine(weights, 4.0, 8)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    shard_dir = tmp_path / "output" / "g" / "shards"
    shard_dir.mkdir(parents=True)
    stats, names = {"clip_id": [], "uid": [], "abs_onset": []}, []
    with tarfile.open(shard_dir / "shards-0000.tar", "w") as tf:
        for r in range(2):
            x = synth.synth_audio(dur + 0.1, seed=500 + r)[:n]
            pcm = np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
            (tmp_path / "audio" / f"rec{r}.flac").write_bytes(FL.encode(pcm.astype(np.int64), 16000))
            names.append(str(tmp_path / "audio" / f"rec{r}.flac"))
            wav = torch.from_numpy(O.pcm_to_mono_float(pcm)).cuda()
            full = eng.infer_waveform(wav, batch_size=8)
            for k in range(3):
                a, b = round(cuts[k] * 16000), min(round(cuts[k + 1] * 16000) + 320, n)   # + the last frame's receptive field
                part = eng.infer_waveform(wav[a:b].contiguous(), batch_size=8)
                rows = (b - a - 320) // step * T
                assert part.shape[0] == rows and torch.equal(part, full[a // 320: a // 320 + rows]), (r, k)
                blob = FL.encode(pcm[a:b].astype(np.int64), 16000)
                info = tarfile.TarInfo(f"rec{r}_{k:04d}.flac")
                info.size = len(blob)
                tf.addfile(info, io.BytesIO(blob))
                stats["clip_id"].append(f"rec{r}_{k:04d}")
                stats["uid"].append(f"rec{r}")
                stats["abs_onset"].append(a / 16000.0)
    eng.close()
    (tmp_path / "manifests" / "g.csv").write_text("path\n" + "\n".join(names) + "\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    common = dict(config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=8)
    cli.main("g", **common)
    clips.main("g", **common)
    out = tmp_path / "output" / "g"
    (out / "stats").mkdir()
    pq.write_table(pa.table(stats), out / "stats" / "clip_stats.parquet")
    full_raw = pq.read_table(out / "vtc_raw" / "shard_0.parquet")
    clip_raw = A._to_absolute(pq.read_table(out / "vtc_clips" / "vtc_raw" / "shard_0.parquet"), pa.table(stats))
    assert full_raw.num_rows > 20
    cov = A.compute_coverage(full_raw, clip_raw)
    for k, c in cov.items():                              # raw runs: same active frames -> same covered time
        assert c["only_full_h"] * 3600 < 1e-6 and c["only_clip_h"] * 3600 < 1e-6, (k, c)
    assert cov["all_speech"]["both_h"] * 3600 > 5.0
    A.main("g")                                           # merged tables: only collar effects at the cuts may differ
    rows = {r["label"]: r for r in __import__("csv").DictReader(open(out / "vtc_clip_alignment_coverage.csv"))}
    assert float(rows["all_speech"]["agreement_pct"]) >= 95.0, rows["all_speech"]
    m = {r["label"]: r for r in __import__("csv").DictReader(open(out / "vtc_clip_alignment_matches.csv"))}
    assert sum(int(r["n_matched"]) for r in m.values()) > 0


def test_launcher_two_workers_shard_files_and_match_one_process(tmp_path, monkeypatch):
    """one worker process per GPU, contiguous equal-count file shards, no collective.  Rehearsed with two
    workers on the one GPU of the test box (--devices 0,0): shard_0 / shard_1 hold disjoint halves of the manifest and
    together equal the single-process run; a rerun resumes from the union of both shards (nothing recomputed)."""
    import os, subprocess, sys
    from pathlib import Path
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    names = []
    for k in range(7):
        x = synth.synth_audio(5.0 + 2.3 * k, seed=700 + k)
        from vtc_amd.audio import write_wav
        write_wav(tmp_path / "audio" / f"r{k}.wav", x)
        names.append(str(tmp_path / "audio" / f"r{k}.wav"))
    for ds in ("two", "one"):
        (tmp_path / "manifests" / f"{ds}.csv").write_text("path\n" + "\n".join(names) + "\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    args = ["--config", str(tmp_path / "cfg.yml"), "--checkpoint", "synthetic:0", "--batch_size", "8"]
    env = dict(os.environ, PYTHONPATH=str(Path(__file__).resolve().parents[1]))
    cmd = [sys.executable, "-m", "vtc_amd.launcher", "two", "--gpus", "2", "--devices", "0,0", "--"] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    cli.main("one", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=8)
    out2, out1 = tmp_path / "output" / "two", tmp_path / "output" / "one"
    m0 = pq.read_table(out2 / "vtc_meta" / "shard_0.parquet").column("uid").to_pylist()
    m1 = pq.read_table(out2 / "vtc_meta" / "shard_1.parquet").column("uid").to_pylist()
    assert m0 == ["r0", "r1", "r2", "r3"] and m1 == ["r4", "r5", "r6"]          # shard_list: sizes 4, 3, contiguous
    key = lambda r: (r["uid"], r["label"], r["onset"])
    for kind in ("vtc_raw", "vtc_merged", "vtc_meta"):
        both = sorted(pq.read_table(out2 / kind / "shard_0.parquet").to_pylist() +
                      pq.read_table(out2 / kind / "shard_1.parquet").to_pylist(), key=lambda r: key(r) if "label" in r else r["uid"])
        single = sorted(pq.read_table(out1 / kind / "shard_0.parquet").to_pylist(), key=lambda r: key(r) if "label" in r else r["uid"])
        assert both == single, kind
    # --balance: size-balanced shards (longest-processing-time) instead of equal counts; still a partition, same rows
    (tmp_path / "manifests" / "bal.csv").write_text("path\n" + "\n".join(names) + "\n")
    rb = subprocess.run([sys.executable, "-m", "vtc_amd.launcher", "bal", "--devices", "0,0", "--"] + args + ["--balance"],
                        env=env, capture_output=True, text=True, timeout=170)
    assert rb.returncode == 0, rb.stdout[-2000:] + rb.stderr[-2000:]
    outb = tmp_path / "output" / "bal"
    b0 = pq.read_table(outb / "vtc_meta" / "shard_0.parquet").column("uid").to_pylist()
    b1 = pq.read_table(outb / "vtc_meta" / "shard_1.parquet").column("uid").to_pylist()
    assert sorted(b0 + b1) == [f"r{k}" for k in range(7)] and not set(b0) & set(b1)
    size = {f"r{k}": os.path.getsize(names[k]) for k in range(7)}
    assert abs(sum(size[u] for u in b0) - sum(size[u] for u in b1)) <= max(size.values())       # LPT bound
    rows_b = sorted(pq.read_table(outb / "vtc_raw" / "shard_0.parquet").to_pylist() +
                    pq.read_table(outb / "vtc_raw" / "shard_1.parquet").to_pylist(), key=key)
    assert rows_b == sorted(pq.read_table(out1 / "vtc_raw" / "shard_0.parquet").to_pylist(), key=key)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "0 remaining" in (r.stdout + r.stderr)
    assert sorted(pq.read_table(out2 / "vtc_meta" / "shard_0.parquet").column("uid").to_pylist()) == ["r0", "r1", "r2", "r3"]


def test_cli_skewed_manifest_mixed_lengths(weights, tmp_path, monkeypatch):
    """BASELINE config C5 in small: durations log-uniform from 10 ms (shorter than the receptive field: error row) to
    minutes, so short files are grouped, long ones run alone, and batches of both kinds alternate in the pipelined GPU
    thread.  Every file's rows must equal the rows of that file run alone through the engine (windows are independent:
    bit-exact), in manifest order, with one meta row per file."""
    from vtc_amd.audio import write_wav
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    rng = np.random.default_rng(55)
    durs = np.exp(rng.uniform(np.log(0.01), np.log(150.0), 36))
    durs[[3, 17]] = [0.02, 0.0249]                                  # < 400 samples
    durs[[5, 6, 7]] = [3.98, 4.0, 4.02]                             # around one window
    durs[20] = 260.0                                                # many batches on its own
    names = []
    for k, d in enumerate(durs):
        write_wav(tmp_path / "audio" / f"s{k:02d}.wav", synth.synth_audio(float(d), seed=900 + k))
        names.append(str(tmp_path / "audio" / f"s{k:02d}.wav"))
    (tmp_path / "manifests" / "skew.csv").write_text("path\n" + "\n".join(names) + "\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    cli.main("skew", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=8, prefetch=3)
    out = tmp_path / "output" / "skew"
    seg = pq.read_table(out / "vtc_raw" / "shard_0.parquet").to_pylist()
    meta = pq.read_table(out / "vtc_meta" / "shard_0.parquet").to_pylist()
    assert [r["uid"] for r in meta] == [f"s{k:02d}" for k in range(len(durs))]               # manifest order, one row each
    eng = _engine(weights, 4.0, 8)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    from scipy.io import wavfile
    n_err = 0
    for k, row in enumerate(meta):
        uid = row["uid"]
        _, pcm = wavfile.read(names[k])
        mine = [(r["onset"], r["offset"], r["duration"], r["label"]) for r in seg if r["uid"] == uid]
        if len(pcm) < 400:
            n_err += 1
            assert "too short" in row["error"] and not mine and row["vtc_n_segments"] == 0
            continue
        assert row["error"] == ""
        logits = eng.infer_waveform(torch.from_numpy(O.pcm_to_mono_float(pcm)).cuda(), batch_size=8).contiguous()
        s_, e_, l_, mx, mean = eng.decode(logits, lo, hi)
        want = O.intervals_to_segments([(int(a), int(b), O.DEFAULT_LABELS[int(c)]) for a, b, c in zip(s_, e_, l_)], uid)
        assert mine == [(r["onset"], r["offset"], r["duration"], r["label"]) for r in want], uid
        ref_row = O.vtc_meta_row(uid, 0.5, want, round(float(torch.sigmoid(torch.tensor(mx, dtype=torch.float32))), 4), round(mean, 4))
        for key in ("vtc_speech_dur", "vtc_n_segments", "vtc_label_counts", "vtc_max_sigmoid", "vtc_mean_sigmoid"):
            assert row[key] == ref_row[key], (uid, key)
    assert n_err >= 2
    eng.close()


def test_cli_bundled_staging_equals_per_file_staging(tmp_path, monkeypatch):
    """Runs of short files are staged as bundles (one pinned buffer, one H2D copy, one front-end launch per bundle:
    pipeline/vtc.py::stage_bundle).  With files that cannot join a bundle in the middle of the run (44.1 kHz stereo WAV,
    FLAC, float32 WAV, a file shorter than the receptive field, a missing file) all three tables must equal those of the
    same manifest staged file by file (VTC_NO_BUNDLE=1), and a batch of 4 windows forces bundles to be cut short."""
    from scipy.io import wavfile
    from oracle import flac_codec as FL
    from vtc_amd.audio import write_wav
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    rng = np.random.default_rng(77)
    names = []
    for k in range(70):
        d = float(np.exp(rng.uniform(np.log(0.5), np.log(20.0))))
        x = synth.synth_audio(d, seed=1200 + k)
        p = tmp_path / "audio" / f"b{k:02d}.wav"
        if k == 9:       # 44.1 kHz stereo
            up = np.interp(np.arange(int(d * 44100)) / 44100.0, np.arange(len(x)) / 16000.0, x).astype(np.float32)
            wavfile.write(p, 44100, np.clip(np.round(np.stack([up, 0.3 * up], axis=1) * 32768), -32768, 32767).astype(np.int16))
        elif k == 21:    # FLAC
            p = p.with_suffix(".flac")
            p.write_bytes(FL.encode(np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int64), 16000))
        elif k == 33:    # float32 WAV
            wavfile.write(p, 16000, x.astype(np.float32))
        elif k == 40:    # shorter than the receptive field
            write_wav(p, x[:399])
        elif k == 41:    # exactly the receptive field: one frame
            write_wav(p, x[:400])
        else:
            write_wav(p, x)
        names.append(str(p))
    names.insert(50, str(tmp_path / "audio" / "missing.wav"))
    for ds in ("bun", "one"):
        (tmp_path / "manifests" / f"{ds}.csv").write_text("path\n" + "\n".join(names) + "\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    staged = []
    real = cli.stage_bundle
    monkeypatch.setattr(cli, "stage_bundle", lambda paths, eng: (staged.append(len(paths)), real(paths, eng))[1])
    cli.main("bun", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=16, prefetch=3)
    assert len(staged) >= 5 and max(staged) >= 4 and sum(staged) >= 60          # bundles really formed
    monkeypatch.setenv("VTC_NO_BUNDLE", "1")
    n_before = len(staged)
    cli.main("one", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=16, prefetch=3)
    assert len(staged) == n_before
    for kind in ("vtc_meta", "vtc_raw", "vtc_merged"):
        a = pq.read_table(tmp_path / "output" / "bun" / kind / "shard_0.parquet")
        b = pq.read_table(tmp_path / "output" / "one" / kind / "shard_0.parquet")
        assert a.num_rows == b.num_rows and a.num_rows > 0
        assert json.dumps(a.to_pylist(), sort_keys=True) == json.dumps(b.to_pylist(), sort_keys=True), kind
    meta = pq.read_table(tmp_path / "output" / "bun" / "vtc_meta" / "shard_0.parquet").to_pylist()
    assert [r["uid"] for r in meta] == [Path(n).stem for n in names]
    errs = {r["uid"]: r["error"] for r in meta if r["error"]}
    assert set(errs) == {"b40", "missing"} and "too short" in errs["b40"]


def test_cli_window_packing_equals_whole_recordings_per_forward(tmp_path, monkeypatch):
    """Window packing (pipeline/vtc.py::pack_forward): recordings share forwards window by window, so a recording may be cut
    over two or more forwards.  All three tables and the saved logits must equal those of the same manifest run with whole
    recordings per forward (VTC_NO_PACK=1, round 2's grouping): windows are independent, so the rows are bit-identical.
    The manifest holds recordings of just under one batch (the 61-of-64 case in small: 15 windows of a 16-window batch),
    of several batches, one long enough to run alone, short ones, a too-short and a missing file."""
    from vtc_amd.audio import write_wav
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    rng = np.random.default_rng(91)
    durs = [58.0] * 6 + [100.0, 3.0, 58.0, 0.02, 1.0, 230.0, 58.0, 58.0, 600.0, 7.5, 58.0, 58.0, 12.0, 58.0, 101.3, 58.0, 58.0, 4.0]
    names = []
    for k, d in enumerate(durs):
        p = tmp_path / "audio" / f"p{k:02d}.wav"
        write_wav(p, synth.synth_audio(float(d) + float(rng.uniform(0, 0.5)) * (d > 1), seed=1500 + k))
        names.append(str(p))
    names.insert(11, str(tmp_path / "audio" / "missing.wav"))
    for ds in ("pk", "np"):
        (tmp_path / "manifests" / f"{ds}.csv").write_text("path\n" + "\n".join(names) + "\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    cli.main("pk", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=16, prefetch=3, save_logits=True)
    stats = dict(cli.PACK_STATS)
    step = 199 * 320
    packed_windows = sum(-(-(int(d * 16000) - 399) // step) for d in durs if 0.03 < d < 500)
    assert stats["forwards"] > 0 and abs(stats["windows"] - packed_windows) <= len(durs)      # every short recording went through it
    print(f"packed forwards: {stats}")
    assert stats["windows"] / stats["forwards"] > 8.0     # mostly full forwards of 16 (a flush whenever the readers lag is allowed)
    monkeypatch.setenv("VTC_NO_PACK", "1")
    cli.main("np", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=16, prefetch=3, save_logits=True)
    assert cli.PACK_STATS["forwards"] == 0
    for kind in ("vtc_meta", "vtc_raw", "vtc_merged"):
        a = pq.read_table(tmp_path / "output" / "pk" / kind / "shard_0.parquet")
        b = pq.read_table(tmp_path / "output" / "np" / kind / "shard_0.parquet")
        assert a.num_rows == b.num_rows and a.num_rows > 0
        assert json.dumps(a.to_pylist(), sort_keys=True) == json.dumps(b.to_pylist(), sort_keys=True), kind
    meta = pq.read_table(tmp_path / "output" / "pk" / "vtc_meta" / "shard_0.parquet").to_pylist()
    assert [r["uid"] for r in meta] == [Path(n).stem for n in names]
    assert {r["uid"] for r in meta if r["error"]} == {"p09", "missing"}
    for uid in ("p00", "p06", "p11", "p14", "p23"):
        a = torch.load(tmp_path / "output" / "pk" / "logits" / f"{uid}-logits_dict_t.pt", weights_only=True)
        b = torch.load(tmp_path / "output" / "np" / "logits" / f"{uid}-logits_dict_t.pt", weights_only=True)
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a), uid
    # a packed forward that fails (here: the 3rd and the 7th raise before anything is enqueued) sends every recording it
    # touched, cut ones included, through the single-recording path: same tables, no error rows beyond the two expected
    monkeypatch.delenv("VTC_NO_PACK")
    (tmp_path / "manifests" / "pf.csv").write_text("path\n" + "\n".join(names) + "\n")
    calls = {"n": 0}
    real = cli.pack_forward

    def flaky(engine, carry, n_windows):
        calls["n"] += 1
        if calls["n"] in (3, 7):
            raise RuntimeError("injected failure of a packed forward")
        return real(engine, carry, n_windows)

    monkeypatch.setattr(cli, "pack_forward", flaky)
    cli.main("pf", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=16, prefetch=3)
    assert calls["n"] > 7
    for kind in ("vtc_meta", "vtc_raw", "vtc_merged"):
        a = pq.read_table(tmp_path / "output" / "pf" / kind / "shard_0.parquet")
        b = pq.read_table(tmp_path / "output" / "np" / kind / "shard_0.parquet")
        assert json.dumps(a.to_pylist(), sort_keys=True) == json.dumps(b.to_pylist(), sort_keys=True), kind


def test_cli_reruns_files_that_overflow_fp16_with_bf16_storage(weights, tmp_path, monkeypatch):
    """A checkpoint whose activations pass 65,504 gives NaN logits in fp16 storage.  The CLI runs such a file again with a
    bf16 engine on the same weights instead of losing it (pipeline/vtc.py: `rescue`): the rows must be exactly those of a
    run configured for bf16, and with engine.bf16_rescue: false the files are error rows as before."""
    from vtc_amd.audio import write_wav
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    w = {k: np.asarray(v) for k, v in weights.items()}
    key = "encoder.layers.5.feed_forward.intermediate_dense.weight"
    w[key] = w[key] * 1.0e5                      # the FFN hidden state of layer 5 leaves the fp16 range for any input
    np.savez(tmp_path / "hot.npz", **w)
    names = []
    for k, d in enumerate((2.5, 9.7, 0.8, 21.3)):
        p = tmp_path / "audio" / f"h{k}.wav"
        write_wav(p, synth.synth_audio(d, seed=1500 + k))
        names.append(str(p))
    for ds in ("resc", "bf", "off"):
        (tmp_path / "manifests" / f"{ds}.csv").write_text("path\n" + "\n".join(names) + "\n")
    base = "model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n"
    (tmp_path / "f16.yml").write_text(base)
    (tmp_path / "bf16.yml").write_text(base + "engine:\n  mfma_dtype: bf16\n")
    (tmp_path / "off.yml").write_text(base + "engine:\n  bf16_rescue: false\n")
    common = dict(checkpoint=str(tmp_path / "hot.npz"), batch_size=8, save_logits=True)
    cli.main("resc", config=str(tmp_path / "f16.yml"), **common)
    cli.main("bf", config=str(tmp_path / "bf16.yml"), **common)
    cli.main("off", config=str(tmp_path / "off.yml"), **common)
    for kind in ("vtc_meta", "vtc_raw", "vtc_merged"):
        a = pq.read_table(tmp_path / "output" / "resc" / kind / "shard_0.parquet")
        b = pq.read_table(tmp_path / "output" / "bf" / kind / "shard_0.parquet")
        assert json.dumps(a.to_pylist(), sort_keys=True) == json.dumps(b.to_pylist(), sort_keys=True), kind
    meta = pq.read_table(tmp_path / "output" / "resc" / "vtc_meta" / "shard_0.parquet").to_pylist()
    assert [r["uid"] for r in meta] == ["h0", "h1", "h2", "h3"] and not any(r["error"] for r in meta)
    for k in range(4):       # the saved logits are the bf16 engine's, finite
        a = torch.load(tmp_path / "output" / "resc" / "logits" / f"h{k}-logits_dict_t.pt", weights_only=True)
        b = torch.load(tmp_path / "output" / "bf" / "logits" / f"h{k}-logits_dict_t.pt", weights_only=True)
        assert all(torch.isfinite(a[l]).all() and torch.equal(a[l], b[l]) for l in a)
    rec = [json.loads(l) for l in (tmp_path / "logs" / "benchmarks.jsonl").read_text().strip().splitlines()]
    assert [r["extra"]["bf16_rescued_files"] for r in rec] == [4, 0, 0]
    off = pq.read_table(tmp_path / "output" / "off" / "vtc_meta" / "shard_0.parquet").to_pylist()
    assert all("non-finite" in (r["error"] or "") for r in off)


def pa_buf(b):
    import pyarrow as pa
    return pa.BufferReader(b)
