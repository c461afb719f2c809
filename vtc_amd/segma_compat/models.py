"""segma.models stand-in: `Models[name].load_from_checkpoint(...)` (reference src/pipeline/vtc.py:185-192).

The returned object owns the weights on the host until `.to(device)`; the GPU engine (libvtc_amd.so) is
created lazily for the (chunk_duration_s, batch_size) the first apply_model_on_audio call asks for.
"""

from __future__ import annotations

from dataclasses import dataclass
from math import prod
from pathlib import Path

import numpy as np
import torch

from ..engine import VtcEngine
from .. import synth

_ANCHOR = "feature_extractor.conv_layers.0.conv.weight"


@dataclass(frozen=True)
class ConvSettings:
    """What `model.conv_settings` carries (reference src/pipeline/vtc.py:194; geometry from
    src/analysis/chunk_artifact.py:43-57)."""
    kernels: tuple = (10, 3, 3, 3, 3, 2, 2)
    strides: tuple = (5, 2, 2, 2, 2, 2, 2)
    paddings: tuple = (0, 0, 0, 0, 0, 0, 0)

    @property
    def rf_step(self) -> int:
        return prod(self.strides)

    @property
    def rf_size(self) -> int:
        return 1 + sum((k - 1) * prod(self.strides[:i]) for i, k in enumerate(self.kernels))

    def n_windows(self, n_samples: int) -> int:
        """Number of output frames for n_samples input samples."""
        return 0 if n_samples < self.rf_size else (n_samples - self.rf_size) // self.rf_step + 1


def _load_state_dict(path: Path) -> dict:
    suf = path.suffix.lower()
    if suf == ".safetensors":
        from safetensors.numpy import load_file
        return dict(load_file(str(path)))
    if suf == ".npz":
        with np.load(str(path), allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    obj = torch.load(str(path), map_location="cpu", weights_only=True)  # never unpickles code
    if isinstance(obj, dict) and "state_dict" in obj:
        obj = obj["state_dict"]   # Lightning layout
    return dict(obj)


# Other public HuBERT-base layouts, mapped onto the HF names the engine reads (the layout of the real VTC-2.0 checkpoint
# is unpinned: segma is not in the reference tree).  (pattern, replacement) pairs applied in order to every key.
_TORCHAUDIO_RULES = (                                 # torchaudio.models.wav2vec2 (hubert_base)
    (r"^encoder\.feature_projection\.", "feature_projection."),
    (r"^encoder\.transformer\.", "encoder."),
)
_FAIRSEQ_RULES = (                                    # fairseq HubertModel
    (r"^feature_extractor\.conv_layers\.(\d+)\.0\.", r"feature_extractor.conv_layers.\1.conv."),
    (r"^feature_extractor\.conv_layers\.0\.2\.", "feature_extractor.conv_layers.0.layer_norm."),
    (r"^layer_norm\.", "feature_projection.layer_norm."),
    (r"^post_extract_proj\.", "feature_projection.projection."),
    (r"^encoder\.pos_conv\.0\.", "encoder.pos_conv_embed.conv."),
    (r"^encoder\.layers\.(\d+)\.self_attn\.", r"encoder.layers.\1.attention."),
    (r"^encoder\.layers\.(\d+)\.self_attn_layer_norm\.", r"encoder.layers.\1.layer_norm."),
    (r"^encoder\.layers\.(\d+)\.fc1\.", r"encoder.layers.\1.feed_forward.intermediate_dense."),
    (r"^encoder\.layers\.(\d+)\.fc2\.", r"encoder.layers.\1.feed_forward.output_dense."),
)
_ANCHORS = (_ANCHOR, "feature_extractor.conv_layers.0.0.weight")


def _rename(sd: dict, rules) -> dict:
    import re
    out = {}
    for k, v in sd.items():
        for pat, rep in rules:
            k2 = re.sub(pat, rep, k)
            if k2 != k:
                k = k2
                break
        out[k] = v
    return out


def normalise_state_dict(sd: dict, n_labels: int) -> dict:
    """Strip whatever wrapper prefix the checkpoint uses, bring torchaudio / fairseq HuBERT names to the HF HubertModel
    names the engine understands, and locate the head (UNPINNED: segma's module names are not in the reference tree)."""
    anchor = [(k, a) for a in _ANCHORS for k in sd if k.endswith(a)]
    if not anchor:
        raise KeyError(f"checkpoint has no '*{_ANCHOR}' tensor; not a HuBERT-style encoder")
    key, a = anchor[0]
    prefix = key[: -len(a)]
    out = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    if a != _ANCHOR:
        out = _rename(out, _FAIRSEQ_RULES)
    elif any(k.startswith("encoder.transformer.") for k in out):
        out = _rename(out, _TORCHAUDIO_RULES)
    if "head.weight" not in out:
        cands = [k for k, v in sd.items() if k.endswith("weight") and tuple(np.shape(v)) == (n_labels, 768)]
        if len(cands) != 1:
            raise KeyError("cannot identify the multi-label head ([n_labels, 768] weight) in the checkpoint")
        out["head.weight"] = sd[cands[0]]
        out["head.bias"] = sd[cands[0][: -len("weight")] + "bias"]
    return out


class VtcModel:
    conv_settings = ConvSettings()

    def __init__(self, state_dict: dict, label_encoder, config):
        self.state_dict_host = state_dict
        self.label_encoder = label_encoder
        self.config = config
        self.device = torch.device("cpu")
        self._engines = {}
        self._rescue = {}

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, label_encoder, config, train: bool = False, **_):
        if train:
            raise NotImplementedError("vtc_amd is inference-only")
        cp = str(checkpoint_path)
        if cp.startswith("synthetic"):
            seed = int(cp.split(":", 1)[1]) if ":" in cp else int(getattr(config.engine, "synthetic_seed", 0))
            sd = synth.make_weights(seed=seed, n_labels=label_encoder.n_labels)
        else:
            sd = normalise_state_dict(_load_state_dict(Path(cp)), label_encoder.n_labels)
        return cls(sd, label_encoder, config)

    def eval(self):
        return self

    def to(self, device):
        self.device = torch.device(device)
        return self

    def engine(self, chunk_duration_s: float, batch_size: int, device=None) -> VtcEngine:
        dev = torch.device(device) if device is not None else self.device
        if dev.type != "cuda":
            raise RuntimeError(f"vtc_amd runs on ROCm GPUs only (requested device '{dev}'); there is no CPU path")
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        max_batch = int(getattr(self.config.engine, "max_batch", 0) or batch_size)
        key = (round(float(chunk_duration_s), 6), max(max_batch, int(batch_size)), index)
        if key not in self._engines:
            for e in self._engines.values():
                e.close()
            self._engines.clear()
            self._engines[key] = VtcEngine(self.state_dict_host, self.label_encoder.n_labels, chunk_duration_s,
                                           max_batch=key[1], mfma_dtype=self.config.engine.mfma_dtype, device=index)
        return self._engines[key]

    def rescue_engine(self, chunk_duration_s: float, batch_size: int, device=None) -> VtcEngine:
        """A second engine on the same weights with bf16 storage (8 exponent bits: no overflow at 65,504), for the rare
        recording whose activations leave the fp16 range (pipeline/vtc.py).  Small batches: it runs one file at a time
        beside the main engine.  Kept apart from engine()'s cache, which holds one engine per geometry."""
        dev = torch.device(device) if device is not None else self.device
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        key = ("bf16", round(float(chunk_duration_s), 6), int(batch_size), index)
        if key not in self._rescue:
            for e in self._rescue.values():
                e.close()
            self._rescue.clear()
            self._rescue[key] = VtcEngine(self.state_dict_host, self.label_encoder.n_labels, chunk_duration_s,
                                          max_batch=int(batch_size), mfma_dtype="bf16", device=index)
        return self._rescue[key]

    def close_rescue(self):
        for e in self._rescue.values():
            e.close()
        self._rescue.clear()

    def any_engine(self):
        return next(iter(self._engines.values()), None)


class _Registry(dict):
    def __missing__(self, name):
        return VtcModel   # every model name in a segma config maps onto the one encoder this build implements


Models = _Registry(surgical_hydra=VtcModel, hydra=VtcModel)
