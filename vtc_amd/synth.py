"""Synthetic inputs for benchmarks and tests: seeded 16 kHz mono audio and seeded
BabyHuBERT-shaped weights.  The real VTC-2.0 checkpoint (git-lfs submodule, reference .gitmodules:1-3)
is not available offline; a real state dict can be passed to VtcEngine instead of these.
"""

from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16_000
CONV_KERNELS = (10, 3, 3, 3, 3, 2, 2)   # reference: src/analysis/chunk_artifact.py:43
LABELS = ("FEM", "MAL", "KCHI", "OCH")  # reference: src/core/__init__.py:7


def synth_audio(duration_s: float, seed: int, dtype=np.float32) -> np.ndarray:
    """3-6 amplitude-modulated harmonic voices (f0 100-400 Hz, random 0.2-3 s on/off) plus
    low-passed noise at about -30 dBFS, so that logits cross zero often."""
    rng = np.random.default_rng(seed)
    n = int(round(duration_s * SAMPLE_RATE))
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    out = np.zeros(n, dtype=np.float64)
    for _ in range(int(rng.integers(3, 7))):
        f0 = rng.uniform(100.0, 400.0)
        vib = 1.0 + 0.01 * np.sin(2 * np.pi * rng.uniform(3.0, 7.0) * t + rng.uniform(0, 2 * np.pi))
        phase = 2 * np.pi * f0 * np.cumsum(vib) / SAMPLE_RATE
        voice = np.zeros(n)
        for h in range(1, 6):
            voice += np.sin(h * phase + rng.uniform(0, 2 * np.pi)) / h
        # on/off gate with 10 ms ramps
        gate = np.zeros(n)
        pos, on = 0, bool(rng.integers(0, 2))
        while pos < n:
            seg = int(rng.uniform(0.2, 3.0) * SAMPLE_RATE)
            if on:
                gate[pos: pos + seg] = 1.0
            pos += seg
            on = not on
        ramp = np.ones(160) / 160.0
        gate = np.convolve(gate, ramp, mode="same")
        am = 0.6 + 0.4 * np.sin(2 * np.pi * rng.uniform(3.0, 8.0) * t + rng.uniform(0, 2 * np.pi))
        out += rng.uniform(0.3, 1.0) * voice * gate * am
    peak = np.abs(out).max()
    if peak > 0:
        out *= 0.5 / peak
    noise = rng.standard_normal(n)
    # one-pole low-pass -> "pink-ish"
    from scipy.signal import lfilter
    noise = lfilter([1.0], [1.0, -0.95], noise)
    noise *= 10 ** (-30 / 20) / (np.sqrt(np.mean(noise ** 2)) + 1e-12)
    return (out + noise).astype(dtype)


def make_weights(seed: int = 0, n_labels: int = 4, logit_std: float = 2.0, family: str = "stress",
                 calibrated: bool = False, outlier_scale: float = 1.0) -> dict:
    """Seeded random weights under HF HubertModel state-dict names (+ head.weight / head.bias).
    numpy PCG64 -> identical tensors on every machine.

    family "stress": every scale is 1/sqrt(fan_in), LayerNorm affine 1 +- 0.1 / +- 0.1, so the attention and FFN branches
        carry as much signal as the residual stream (a harder numerical case than a HuBERT init: deep random post-LN
        stacks of this kind collapse the frames onto one direction, so the logits move little around a common offset).
    family "hubert": the HuBERT-base initialisation SURVEY.md 8(d) prescribes (transformers HubertPreTrainedModel.
        _init_weights with initializer_range 0.02: Linear N(0, 0.02) / bias 0, LayerNorm and GroupNorm 1 / 0, Conv1d
        kaiming-normal, weight-norm g = ||v||).
    family "trained": the "hubert" draws, then statistics a trained post-LN encoder shows and an initialisation does not
        (_trained_statistics below): six residual channels that sit at +-(50..100) x outlier_scale in every LayerNorm output
        while the other channels stay at unit scale ("massive activations": near-constant, carried by the LayerNorm biases),
        LayerNorm / GroupNorm gains far from 1 (the gains behind an outlier-dominated row undo its small rstd: ~6.7 x a
        log-normal spread; 0.1 on the outlier channels), non-zero biases everywhere, and small read weights on the outlier
        channels.  This is the case that probes the 16-bit residual rows (precision next to large values, 65,504).
    calibrated: (family "trained": first subtract the committed per-label mean logit HEAD_CENTER from head.bias, so that every
        label crosses the threshold; then) scale head.weight and head.bias by the committed factor (HEAD_CALIBRATION, made by tests/golden/make_head_calibration.py
        with the CPU oracle) that brings the oracle's logit std on the calibration clip to `logit_std` = 2 (SURVEY.md 8d:
        "head scaled so logit std ~ 2")."""
    if family not in ("stress", "hubert", "trained"):
        raise ValueError(f"unknown weight family {family!r}")
    rng = np.random.default_rng(seed)
    hub = family in ("hubert", "trained")

    def randn(*shape, std=1.0):
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(std)

    def affine(n):    # LayerNorm / GroupNorm weight, bias (the draws are made in both families: same stream positions)
        a, b = randn(n, std=0.1), randn(n, std=0.1)
        return (np.ones(n, np.float32), np.zeros(n, np.float32)) if hub else (1.0 + a, b)

    def bias(n, std=0.02):
        b = randn(n, std=std)
        return np.zeros(n, np.float32) if hub else b

    def lin(n, k):
        return randn(n, k, std=0.02 if hub else k ** -0.5)

    w = {}
    cin = 1
    for i, k in enumerate(CONV_KERNELS):
        w[f"feature_extractor.conv_layers.{i}.conv.weight"] = randn(512, cin, k, std=(2.0 / (cin * k)) ** 0.5)
        cin = 512
    (w["feature_extractor.conv_layers.0.layer_norm.weight"],
     w["feature_extractor.conv_layers.0.layer_norm.bias"]) = affine(512)
    w["feature_projection.layer_norm.weight"], w["feature_projection.layer_norm.bias"] = affine(512)
    w["feature_projection.projection.weight"] = lin(768, 512)
    w["feature_projection.projection.bias"] = bias(768)
    v = randn(768, 48, 128, std=(2.0 if hub else 1.0) ** 0.5 * (48 * 128) ** -0.5)
    w["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = v
    g = randn(1, 1, 128, std=0.1)
    w["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (
        np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
        * (np.float32(1.0) if hub else (1.0 + g)))
    w["encoder.pos_conv_embed.conv.bias"] = bias(768)
    w["encoder.layer_norm.weight"], w["encoder.layer_norm.bias"] = affine(768)
    for l in range(12):
        p = f"encoder.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[p + f"attention.{nm}.weight"] = lin(768, 768)
            w[p + f"attention.{nm}.bias"] = bias(768)
        w[p + "layer_norm.weight"], w[p + "layer_norm.bias"] = affine(768)
        w[p + "feed_forward.intermediate_dense.weight"] = lin(3072, 768)
        w[p + "feed_forward.intermediate_dense.bias"] = bias(3072)
        w[p + "feed_forward.output_dense.weight"] = lin(768, 3072)
        w[p + "feed_forward.output_dense.bias"] = bias(768)
        w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"] = affine(768)
    w["head.weight"] = randn(n_labels, 768, std=logit_std * 768 ** -0.5)
    w["head.bias"] = randn(n_labels, std=0.5)
    if family == "trained":
        _trained_statistics(w, seed, outlier_scale)
    if calibrated:
        key = f"{family}:{seed}"
        if outlier_scale != 1.0:
            key += f":x{outlier_scale:g}"
        if n_labels != 4 or key not in HEAD_CALIBRATION:
            raise ValueError(f"no committed head calibration for {key} with {n_labels} labels")
        scale = np.float32(logit_std / HEAD_CALIBRATION[key])
        if key in HEAD_CENTER:     # per-label mean logit on the calibration clip, removed before scaling (family "trained")
            w["head.bias"] = w["head.bias"] - np.asarray(HEAD_CENTER[key], dtype=np.float32)
        w["head.weight"] = w["head.weight"] * scale
        w["head.bias"] = w["head.bias"] * scale
    return w


N_OUTLIER_CHANNELS = 6


def _trained_statistics(w: dict, seed: int, outlier_scale: float = 1.0) -> None:
    """Turn a "hubert" draw into the "trained" family in place (its own PCG64 stream, so the base tensors are those of
    family="hubert" for the same seed).  Fixed point of the construction: every LayerNorm output x has x_S = beta_S =
    +-(50..100) on the outlier set S and unit-scale values elsewhere; the next pre-LN row u = x + branch then has
    std_u ~ sqrt(sum beta_S^2 / 768) ~ 6.7, so that LayerNorm's gain on the other channels is std_u x log-normal(0.25) and
    restores the unit scale, while its gain on S is 0.1 and its bias on S re-creates the outliers."""
    rng = np.random.default_rng([seed, 0x7A1])
    f32 = np.float32
    S = np.sort(rng.choice(768, N_OUTLIER_CHANNELS, replace=False))
    sign = rng.choice(np.array([-1.0, 1.0]), N_OUTLIER_CHANNELS)

    def spread(n):
        return np.exp(rng.standard_normal(n) * 0.25)

    def outliers():
        return sign * rng.uniform(50.0, 100.0, N_OUTLIER_CHANNELS) * outlier_scale

    for nm in ("feature_extractor.conv_layers.0.layer_norm", "feature_projection.layer_norm"):
        n = w[nm + ".weight"].shape[0]
        w[nm + ".weight"] = spread(n).astype(f32)
        w[nm + ".bias"] = (rng.standard_normal(n) * 0.1).astype(f32)

    def layer_norm(nm, std_in):
        g = std_in * spread(768)
        b = rng.standard_normal(768) * 0.1
        a = outliers()
        g[S], b[S] = 0.1, a
        w[nm + ".weight"], w[nm + ".bias"] = g.astype(f32), b.astype(f32)
        m = a.sum() / 768.0
        return float(np.sqrt((a ** 2).sum() / 768.0 - m * m + 1.0))     # std of the next pre-LN row (unit-scale rest)

    std_u = layer_norm("encoder.layer_norm", 1.0)
    for l in range(12):
        p = f"encoder.layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj"):
            w[p + f"attention.{nm}.weight"][:, S] *= f32(0.2)
        w[p + "feed_forward.intermediate_dense.weight"][:, S] *= f32(0.2)
        for nm, n in (("attention.q_proj", 768), ("attention.k_proj", 768), ("attention.v_proj", 768), ("attention.out_proj", 768),
                      ("feed_forward.intermediate_dense", 3072), ("feed_forward.output_dense", 768)):
            w[p + nm + ".bias"] = (rng.standard_normal(n) * 0.05).astype(f32)
        std_u = layer_norm(p + "layer_norm", std_u)
        std_u = layer_norm(p + "final_layer_norm", std_u)
    w["feature_projection.projection.bias"] = (rng.standard_normal(768) * 0.05).astype(f32)
    w["encoder.pos_conv_embed.conv.bias"] = (rng.standard_normal(768) * 0.05).astype(f32)
    w["head.weight"][:, S] *= f32(0.01 / outlier_scale)


# Oracle logit std of the uncalibrated weights on the calibration clip (synth_audio(40 s, seed 9000), 10 s windows, full
# windows only); written by tests/golden/make_head_calibration.py, which also commits tests/golden/head_calibration.json.
HEAD_CALIBRATION: dict = {"stress:0": 0.864546, "stress:1": 2.370401, "stress:2": 0.574449, "stress:3": 2.047762,
                          "hubert:0": 2.250636, "hubert:1": 2.099057, "hubert:2": 1.722512, "hubert:3": 2.136115,
                          "hubert:4": 2.133864, "hubert:5": 1.794613, "hubert:6": 1.870844, "hubert:7": 2.578898,
                          "trained:0": 2.250929, "trained:1": 2.000487, "trained:2": 2.357368, "trained:3": 2.190693}
# family "trained": per-label mean logit of the uncalibrated weights on that clip (same script)
HEAD_CENTER: dict = {"trained:0": [7.796665, -2.117744, -3.57752, 0.319455], "trained:1": [-3.903499, 6.581756, -2.683167, 2.39855],
                     "trained:2": [-20.389849, -10.928957, 9.099744, 4.988534], "trained:3": [1.127458, -1.794585, 3.962687, 1.665586]}
