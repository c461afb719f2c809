"""CPU oracle for the VTC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``vtc_amd/`` imports it; the product
path fails loudly when the HIP library is missing instead of falling back here.

What it restates (fp32, PyTorch-CPU / numpy), with the evidence each piece follows:

* window tiling / stitching ........ /root/reference/src/analysis/chunk_artifact.py:43-83,
                                     /root/reference/src/packaging/clips.py:48-52
* BabyHuBERT ("HuBERT surgical_hydra") encoder -- the arithmetic of the HuBERT-base
  stack the reference names at chunk_artifact.py:40-48.  segma (the package that owns
  the model code, pyproject.toml:31) is absent offline, so the arithmetic follows the
  locally installed HF implementation, transformers/models/hubert/modeling_hubert.py:
  feature encoder :154-213, projection :216-231, positional conv :45-103, encoder
  :407-442, layer :371-404, attention :234-260, feed-forward :347-368.
* threshold -> intervals ........... call sites src/pipeline/vtc.py:77-96, rule pinned by
                                     chunk_artifact.py:11-13,75-83
* intervals -> segments ............ src/core/intervals.py:85-108
* meta rows ........................ src/core/metadata.py:12-50
* collar merge ..................... src/utils.py:323-390
* shard slicing .................... src/utils.py:271-285

PARITY STATUS.  The encoder restatement is pinned against ``transformers.HubertModel``
(same state dict, <=1e-5, tests/golden/make_golden.py, run in the build container) and the
meta rows against the reference's own ``src/core/metadata.py`` imported there.  Everything
that lives only inside segma (head topology, tail-window rule, input normalisation,
threshold inclusivity, label order) has no golden vector anywhere in the reference:
for those items this oracle is **parity unpinned** (SURVEY.md section 8c).
"""

from __future__ import annotations

import json
import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16_000
CONV_KERNELS = (10, 3, 3, 3, 3, 2, 2)   # chunk_artifact.py:43
CONV_STRIDES = (5, 2, 2, 2, 2, 2, 2)    # chunk_artifact.py:44
RF_STEP = math.prod(CONV_STRIDES)       # 320 samples = 20 ms  (chunk_artifact.py:51)
RF_SIZE = 1 + sum((k - 1) * math.prod(CONV_STRIDES[:i]) for i, k in enumerate(CONV_KERNELS))  # 400
DEFAULT_LABELS = ("FEM", "MAL", "KCHI", "OCH")  # src/core/__init__.py:7


# --------------------------------------------------------------------------- geometry
def conv_out_len(n: int) -> int:
    """Length after the 7 un-padded conv layers (modeling_hubert.py:160-166 geometry)."""
    for k, s in zip(CONV_KERNELS, CONV_STRIDES):
        if n < k:
            return 0
        n = (n - k) // s + 1
    return n


def n_frames_for_samples(n: int) -> int:
    """Closed form of conv_out_len: frame g exists iff g*320 + 400 <= n."""
    return 0 if n < RF_SIZE else (n - RF_SIZE) // RF_STEP + 1


def chunk_artifact_n_windows(chunk_f: int) -> int:
    """The reference's own frame-count formula (chunk_artifact.py:59-69), incl. its +1 divisor."""
    has_even = any(k % 2 == 0 for k in CONV_KERNELS)
    span = sum((k - 1) * math.prod(CONV_STRIDES[:i]) for i, k in enumerate(CONV_KERNELS)) + 1
    return math.floor((chunk_f - span) / (RF_STEP + (1 if has_even else 0))) + 1


def peak_seconds(n_chunks: int, frames_per_chunk: int = 199) -> float:
    """chunk_artifact.py:80-83."""
    rt = sum((1 - k) * math.prod(CONV_STRIDES[:i]) for i, k in enumerate(CONV_KERNELS))  # -399
    return ((n_chunks * frames_per_chunk - 1) * RF_STEP - rt + 1) / SAMPLE_RATE


@dataclass
class WindowPlan:
    window: int                 # samples per window (chunk_duration_s * 16000)
    frames_per_window: int      # T
    step: int                   # T * 320   (chunk_artifact.py:71-73)
    starts: list = field(default_factory=list)   # sample offset of each window
    valid: list = field(default_factory=list)    # real samples available (rest zero padded)
    keep: list = field(default_factory=list)     # frames kept from each window

    @property
    def total_frames(self) -> int:
        return int(sum(self.keep))


def plan_windows(n_samples: int, chunk_duration_s: float) -> WindowPlan:
    """Tile a file into windows stepped by T*320 samples; global frame g <-> sample g*320.

    Tail rule (UNPINNED, segma absent): the last window is zero padded to full length and
    only the frames whose 400-sample receptive field lies inside the real audio are kept,
    so the stitched output has exactly n_frames_for_samples(n_samples) rows.
    """
    window = int(round(chunk_duration_s * SAMPLE_RATE))
    t = conv_out_len(window)
    if t <= 0:
        raise ValueError(f"chunk of {window} samples is shorter than the receptive field")
    plan = WindowPlan(window=window, frames_per_window=t, step=t * RF_STEP)
    s = 0
    while n_samples - s >= RF_SIZE:
        rem = n_samples - s
        plan.starts.append(s)
        plan.valid.append(min(rem, window))
        plan.keep.append(min(t, n_frames_for_samples(rem)))
        s += plan.step
    return plan


# --------------------------------------------------------------------------- weights
def make_weights(seed: int = 0, n_labels: int = 4, logit_std: float = 2.0, family: str = "stress",
                 calibrated: bool = False, outlier_scale: float = 1.0) -> dict:
    """Seeded random BabyHuBERT-shaped weights (the real VTC-2.0 checkpoint is absent offline) as torch tensors.

    The tensors are INPUT DATA shared with the HIP path, so there is one generator: vtc_amd.synth.make_weights (numpy
    PCG64, the same tensors on every machine; families "stress" = 1/sqrt(fan_in) scales, "hubert" = the HuBERT-base
    initialisation of SURVEY.md 8d, "trained" = that plus outlier residual channels and non-unit LayerNorm gains; calibrated=True applies the committed head scale of tests/golden/head_calibration.json).
    Names follow the HF state dict so the same dict can be loaded into transformers.HubertModel for pinning; the
    "hydra" head is one linear unit per label on the last hidden state (UNPINNED topology)."""
    from vtc_amd import synth
    w = synth.make_weights(seed=seed, n_labels=n_labels, logit_std=logit_std, family=family, calibrated=calibrated,
                           outlier_scale=outlier_scale)
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}


def pos_conv_weight(w: dict) -> torch.Tensor:
    """Fold weight_norm(dim=2): W = g * v / ||v||_(dims 0,1)   (modeling_hubert.py:60-78)."""
    v = w["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    g = w["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    return g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()


# --------------------------------------------------------------------------- encoder
def _q(x: torch.Tensor, emulate_bf16: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if emulate_bf16 else x


def encoder_stages(w: dict, wav: torch.Tensor, emulate_bf16: bool = False, ranges: dict | None = None) -> dict:
    """Forward (B, W) fp32 waveforms; returns every stage output (fp32).

    emulate_bf16=True rounds every GEMM operand (activations and weights) to bf16 with fp32
    accumulation, which is what the MFMA path does; used to size the error budget on CPU.
    ranges: if a dict, receives max |value| of every tensor class the HIP path keeps in 16-bit storage between kernels
    (conv activations, pre-LayerNorm residual rows, q/k/v, attention context, FFN hidden): how close a weight set comes
    to the fp16 limit of 65,504.
    """
    q = lambda t: _q(t, emulate_bf16)
    out = {}

    def see(name, t):
        if ranges is not None:
            ranges[name] = max(ranges.get(name, 0.0), float(t.abs().max()))
        return t
    x = wav[:, None, :]
    # conv 0 + GroupNorm(512 groups) + GELU   (modeling_hubert.py:154-175); conv0 stays fp32 in HIP too
    x = F.conv1d(x, w["feature_extractor.conv_layers.0.conv.weight"], stride=CONV_STRIDES[0])
    x = F.group_norm(x, 512, w["feature_extractor.conv_layers.0.layer_norm.weight"],
                     w["feature_extractor.conv_layers.0.layer_norm.bias"], eps=1e-5)
    x = see("conv_activations", F.gelu(x))
    out["conv0"] = x
    for i in range(1, 7):  # modeling_hubert.py:106-124
        x = F.conv1d(q(x), q(w[f"feature_extractor.conv_layers.{i}.conv.weight"]), stride=CONV_STRIDES[i])
        x = see("conv_activations", F.gelu(x))
        out[f"conv{i}"] = x
    x = x.transpose(1, 2)  # (B, T, 512)
    # feature projection (modeling_hubert.py:216-231)
    x = F.layer_norm(q(x), (512,), w["feature_projection.layer_norm.weight"],
                     w["feature_projection.layer_norm.bias"], eps=1e-5)
    x = F.linear(q(x), q(w["feature_projection.projection.weight"]), w["feature_projection.projection.bias"])
    out["proj"] = x
    # positional conv + residual + LN (modeling_hubert.py:83-92, 439-441)
    pc = F.conv1d(q(x.transpose(1, 2)), q(pos_conv_weight(w)), w["encoder.pos_conv_embed.conv.bias"],
                  padding=64, groups=16)[:, :, :-1]
    x = see("pre_layernorm_rows", x + F.gelu(pc).transpose(1, 2))
    x = F.layer_norm(x, (768,), w["encoder.layer_norm.weight"], w["encoder.layer_norm.bias"], eps=1e-5)
    out["enc_in"] = x
    b, t, _ = x.shape
    for l in range(12):  # modeling_hubert.py:387-397 (post-LN)
        p = f"encoder.layers.{l}."
        xq = q(x)
        qh = F.linear(xq, q(w[p + "attention.q_proj.weight"]), w[p + "attention.q_proj.bias"]) * 0.125
        kh = F.linear(xq, q(w[p + "attention.k_proj.weight"]), w[p + "attention.k_proj.bias"])
        vh = F.linear(xq, q(w[p + "attention.v_proj.weight"]), w[p + "attention.v_proj.bias"])
        qh, kh, vh = (q(see("qkv", z)).view(b, t, 12, 64).transpose(1, 2) for z in (qh, kh, vh))
        att = torch.softmax(qh @ kh.transpose(2, 3), dim=-1)   # modeling_hubert.py:248-253
        ctx = see("attention_context", (q(att) @ vh).transpose(1, 2).reshape(b, t, 768))
        x = see("pre_layernorm_rows", x + F.linear(q(ctx), q(w[p + "attention.out_proj.weight"]), w[p + "attention.out_proj.bias"]))
        x = F.layer_norm(x, (768,), w[p + "layer_norm.weight"], w[p + "layer_norm.bias"], eps=1e-5)
        h = see("ffn_hidden", F.gelu(F.linear(q(x), q(w[p + "feed_forward.intermediate_dense.weight"]),
                                              w[p + "feed_forward.intermediate_dense.bias"])))
        x = see("pre_layernorm_rows", x + F.linear(q(h), q(w[p + "feed_forward.output_dense.weight"]),
                                                   w[p + "feed_forward.output_dense.bias"]))
        x = F.layer_norm(x, (768,), w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], eps=1e-5)
        out[f"layer{l}"] = x
    out["logits"] = F.linear(x, w["head.weight"], w["head.bias"])   # head stays fp32
    return out


@torch.no_grad()
def encoder_logits(w: dict, wav: torch.Tensor, emulate_bf16: bool = False, ranges: dict | None = None) -> torch.Tensor:
    return encoder_stages(w, wav, emulate_bf16, ranges)["logits"]


@torch.no_grad()
def apply_model_on_waveform(w: dict, wav: np.ndarray, chunk_duration_s: float, batch_size: int = 8,
                            emulate_bf16: bool = False, ranges: dict | None = None) -> torch.Tensor:
    """Oracle of segma.inference.apply_model_on_audio (call site src/pipeline/vtc.py:236-243)
    after decoding: tile -> batched forward -> concatenate frames -> (n_frames, n_labels)."""
    wav = np.asarray(wav, dtype=np.float32)
    plan = plan_windows(len(wav), chunk_duration_s)
    if not plan.starts:
        raise ValueError(f"audio too short: {len(wav)} samples < {RF_SIZE}")
    rows = []
    for b0 in range(0, len(plan.starts), batch_size):
        idx = range(b0, min(b0 + batch_size, len(plan.starts)))
        batch = np.zeros((len(idx), plan.window), dtype=np.float32)
        for r, i in enumerate(idx):
            batch[r, : plan.valid[i]] = wav[plan.starts[i]: plan.starts[i] + plan.valid[i]]
        lg = encoder_logits(w, torch.from_numpy(batch), emulate_bf16, ranges)
        for r, i in enumerate(idx):
            rows.append(lg[r, : plan.keep[i]])
    return torch.cat(rows, dim=0)


# ----------------------------------------------------------------------------- audio front-end (parity unpinned)
def pcm_to_mono_float(pcm: np.ndarray) -> np.ndarray:
    """int16 / float32 PCM, (n,) or (n, channels) -> float32 mono: int16 / 32768, channels averaged.
    segma's rule is not in the reference: parity unpinned."""
    x = np.asarray(pcm)
    x = x.astype(np.float32) / np.float32(32768.0) if x.dtype == np.int16 else x.astype(np.float32)
    if x.ndim == 2:
        acc = np.zeros(x.shape[0], dtype=np.float32)
        for c in range(x.shape[1]):
            acc = acc + x[:, c]
        x = acc / np.float32(x.shape[1]) if x.shape[1] > 1 else acc
    return x


def resample_sinc_hann(x: np.ndarray, sr_in: int, sr_out: int = SAMPLE_RATE) -> np.ndarray:
    """torchaudio.functional.resample (the library segma pulls in, reference requirements.txt:393-394; absent here),
    restated from its published algorithm: resampling_method="sinc_interp_hann", lowpass_filter_width=6, rolloff=0.99;
    kernel built in float64 and cast to float32, applied as a strided conv1d, output trimmed to ceil(new*n/orig)."""
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    wav = torch.from_numpy(np.asarray(x, dtype=np.float32))
    if orig == new:
        return wav.numpy()
    lowpass_filter_width, rolloff = 6, 0.99
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = t * base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = (kernels * window * scale).to(torch.float32)                     # [new, 1, 2*width + orig]
    length = wav.shape[0]
    padded = F.pad(wav[None, None], (width, width + orig))
    res = F.conv1d(padded, kernels, stride=orig)                               # [1, new, blocks]
    res = res.transpose(1, 2).reshape(-1)
    target = int(math.ceil(new * length / orig))
    return res[:target].contiguous().numpy()


# --------------------------------------------------------------------------- decode
def apply_thresholds(logits: torch.Tensor, thresholds: dict, labels=DEFAULT_LABELS) -> torch.Tensor:
    """sigmoid(logit) within [lower_bound, upper_bound] per label (call site vtc.py:84-90).
    Bounds inclusive (UNPINNED)."""
    p = torch.sigmoid(logits.to(torch.float32))
    lo = torch.tensor([thresholds[l]["lower_bound"] for l in labels], dtype=torch.float32)
    hi = torch.tensor([thresholds[l]["upper_bound"] for l in labels], dtype=torch.float32)
    return ((p >= lo) & (p <= hi)).to(torch.uint8)


def create_intervals(mask, labels=DEFAULT_LABELS) -> list:
    """Run-length encode each label column: run i..j -> (i*320, j*320+400, label)
    (chunk_artifact.py:11-13,75-83).  Order: label-major, then time."""
    m = np.asarray(mask).astype(np.uint8)
    out = []
    for j, lab in enumerate(labels):
        col = np.concatenate([[0], m[:, j], [0]]).astype(np.int8)
        d = np.diff(col)
        starts = np.flatnonzero(d == 1)
        ends = np.flatnonzero(d == -1) - 1
        for s, e in zip(starts, ends):
            out.append((int(s) * RF_STEP, int(e) * RF_STEP + RF_SIZE, lab))
    return out


def frames_to_seconds(sample_index) -> float:
    """'sample-index intervals' -> seconds (src/core/intervals.py:72-75,95-96)."""
    return sample_index / SAMPLE_RATE


def intervals_to_segments(intervals, uid: str) -> list:
    """src/core/intervals.py:85-108."""
    rows = []
    for s, e, lab in intervals:
        on = round(float(frames_to_seconds(s)), 3)
        off = round(float(frames_to_seconds(e)), 3)
        dur = round(off - on, 3)
        if dur > 0:
            rows.append(dict(uid=uid, onset=on, offset=off, duration=dur, label=lab))
    return rows


def sigmoid_summary(logits: torch.Tensor):
    """src/pipeline/vtc.py:266-273."""
    p = logits.to(torch.float32).sigmoid()
    return round(float(p.max().item()), 4), round(float(p.mean().item()), 4)


def vtc_meta_row(uid, threshold, segments, max_sigmoid, mean_sigmoid) -> dict:
    """src/core/metadata.py:28-50."""
    counts, dur = {}, 0.0
    for s in segments:
        dur += s["duration"]
        counts[s["label"]] = counts.get(s["label"], 0) + 1
    return dict(uid=uid, vtc_threshold=threshold, vtc_speech_dur=round(dur, 3), vtc_n_segments=len(segments),
                vtc_label_counts=json.dumps(counts), vtc_max_sigmoid=max_sigmoid,
                vtc_mean_sigmoid=mean_sigmoid, error="")


def vtc_error_row(uid, error) -> dict:
    """src/core/metadata.py:12-25."""
    return dict(vtc_threshold=float("nan"), vtc_speech_dur=0.0, vtc_n_segments=0, vtc_label_counts="{}",
                vtc_max_sigmoid=float("nan"), vtc_mean_sigmoid=float("nan"), error=error, uid=uid)


def merge_segments(rows: list, min_duration_off_s: float = 0.1, min_duration_on_s: float = 0.1) -> list:
    """src/utils.py:323-390 as a plain loop: per (uid,label) sorted by onset, a new group starts
    when onset > cummax(previous offsets) + collar; group -> (min onset, max offset) rounded to
    3 dp; keep duration >= min_on; result sorted by (uid, label, onset)."""
    parts = {}
    for r in rows:
        parts.setdefault((r["uid"], r["label"]), []).append((r["onset"], r["offset"]))
    out = []
    for (uid, lab) in sorted(parts):
        segs = sorted(parts[(uid, lab)], key=lambda p: p[0])
        groups, prev_max = [], 0.0   # fill_null(0.0) for the first row (utils.py:360)
        for i, (on, off) in enumerate(segs):
            if i == 0:
                new = on > 0.0 + min_duration_off_s
                groups.append([on, off])  # first row always opens group 0 whatever _new_grp says
            else:
                new = on > prev_max + min_duration_off_s
                if new:
                    groups.append([on, off])
                else:
                    groups[-1][0] = min(groups[-1][0], on)
                    groups[-1][1] = max(groups[-1][1], off)
            prev_max = off if i == 0 else max(prev_max, off)
        for on, off in groups:
            on, off = round(on, 3), round(off, 3)
            dur = round(off - on, 3)
            if dur >= min_duration_on_s:
                out.append(dict(uid=uid, onset=on, offset=off, duration=dur, label=lab))
    return out


def shard_list(items: list, array_id: int, array_count: int) -> list:
    """src/utils.py:271-285."""
    base, rem = divmod(len(items), array_count)
    start = array_id * base + min(array_id, rem)
    return items[start: start + base + (1 if array_id < rem else 0)]
