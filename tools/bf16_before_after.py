#!/usr/bin/env python3
"""bf16 mode before and after the LayerNorm-free layers: the same C1-style recordings through two builds of
the library, against the fp32 CPU oracle.  `--lib` names a libvtc_amd.so built from an older commit's vtc_amd/csrc (same C
ABI); without it the in-tree library runs.  Dev tool: uses the oracle as the checker.

    git archive <commit> vtc_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/vtc_amd/csrc
    python tools/bf16_before_after.py --lib /tmp/old/vtc_amd/libvtc_amd.so --dtype bf16
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import vtc_oracle as O  # noqa: E402
from oracle.parity_metrics import compare  # noqa: E402
from vtc_amd import lib, synth  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--chunks", type=float, nargs="+", default=[4.0, 10.0])
    ap.add_argument("--family", default="hubert")
    args = ap.parse_args()
    if args.lib:
        lib.LIB_PATH = Path(args.lib).resolve()
    from vtc_amd.engine import VtcEngine
    for seed in args.seeds:
        w = O.make_weights(seed=seed, family=args.family, calibrated=True)
        for chunk in args.chunks:
            audio = synth.synth_audio(args.seconds, seed=4000 + 10 * seed + int(chunk))
            ref = O.apply_model_on_waveform(w, audio, chunk, batch_size=8)
            eng = VtcEngine(w, n_labels=4, chunk_duration_s=chunk, max_batch=16, mfma_dtype=args.dtype, device=0)
            got = eng.infer_waveform(torch.from_numpy(audio).to(eng.device), batch_size=16).cpu()
            eng.close()
            c = compare(ref, got)
            print(json.dumps({"library": args.lib or "in-tree", "dtype": args.dtype, "family": args.family, "weight_seed": seed,
                              "chunk_duration_s": chunk, "max_abs_logit_err": c["max_abs_logit_err"],
                              "mean_abs_logit_err": c["mean_abs_logit_err"], "frame_label_agreement": c["frame_label_agreement"],
                              "flipped_frame_labels": c["flipped_frame_labels"], "frames": c["frames"]}), flush=True)


if __name__ == "__main__":
    main()
