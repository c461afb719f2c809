#!/usr/bin/env python3
"""Parity figures of SURVEY.md 8(d) on a larger sample than the tests use: the HIP path against the fp32 CPU oracle
for several weight families / seeds, window lengths and recordings -- max / mean logit error, frame-label agreement, and
how far segment boundaries move in vtc_raw and in vtc_merged (the shipped table; the 0.3 s collar of the reference's
merge_segments_df, src/utils.py:323-390, re-joins one-frame splits).  Writes one JSON document.

    python tools/parity_report.py --seconds 600 --seeds 0 1 2 3 --family hubert stress --out profiles/r02_parity_report.json
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import vtc_oracle as O  # noqa: E402   (a dev tool may use the oracle as the checker)
from vtc_amd import synth  # noqa: E402
from vtc_amd.engine import VtcEngine  # noqa: E402

from oracle.parity_metrics import THR, boundary_match, compare  # noqa: E402,F401


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3])
    ap.add_argument("--chunks", type=float, nargs="+", default=[4.0, 10.0])
    ap.add_argument("--family", nargs="+", default=["hubert", "stress"])
    ap.add_argument("--uncalibrated", action="store_true")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    cases = []
    for family in args.family:
        for seed in args.seeds:
            if family == "stress" and seed > 3 or family == "trained" and seed > 3:
                continue            # head calibrations are committed for seeds 0-3 of these families
            w = O.make_weights(seed=seed, family=family, calibrated=not args.uncalibrated)
            for chunk in args.chunks:
                eng = VtcEngine(w, n_labels=4, chunk_duration_s=chunk, max_batch=16, mfma_dtype=args.dtype, device=0)
                audio = synth.synth_audio(args.seconds, seed=4000 + 10 * seed + int(chunk))
                t0 = time.time()
                ranges = {}
                ref = O.apply_model_on_waveform(w, audio, chunk, batch_size=8, ranges=ranges)
                t_ref = time.time() - t0
                got = eng.infer_waveform(torch.from_numpy(audio).to(eng.device), batch_size=16)
                torch.cuda.synchronize()
                c = {"family": family, "weight_seed": seed, "head_calibrated": not args.uncalibrated,
                     "chunk_duration_s": chunk, "audio_seconds": len(audio) / 16000.0}
                c.update(compare(ref, got.cpu()))
                c["oracle_seconds"] = round(t_ref, 1)
                c["oracle_max_abs_activation"] = {k: round(v, 2) for k, v in ranges.items()}     # against fp16's 65,504
                cases.append(c)
                print(json.dumps(c), flush=True)
                eng.close()
    doc = {"what": f"HIP path ({args.dtype} MFMA operands, fp32 accumulate) vs fp32 CPU oracle on seeded synthetic audio and seeded "
                   "random BabyHuBERT-base weights (vtc_amd.synth.make_weights families; head calibrated to logit std 2 unless "
                   "stated); tolerances of BASELINE.json north_star: logits <= 2e-2 abs, agreement >= 99.9 %, boundaries within "
                   "one 20 ms frame (fraction of segments with both boundaries within one frame, raw and collar-merged tables)",
           "worst_max_abs_logit_err": max(c["max_abs_logit_err"] for c in cases),
           "worst_agreement": min(c["frame_label_agreement"] for c in cases),
           "worst_raw_boundary_match": min(min(c["raw"]["oracle_matched"], c["raw"]["gpu_matched"]) for c in cases),
           "worst_merged_boundary_match": min(min(c["merged"]["oracle_matched"], c["merged"]["gpu_matched"]) for c in cases),
           "unattributed_boundary_mismatches": sum(c["attribution"][t]["unattributed"] for c in cases for t in ("raw", "merged")),
           "worst_oracle_abs_logit_at_a_flip": max(c["attribution"]["max_oracle_abs_logit_at_flips"] for c in cases),
           "by_family": {f: {"worst_agreement": min(c["frame_label_agreement"] for c in cases if c["family"] == f),
                             "worst_max_abs_logit_err": max(c["max_abs_logit_err"] for c in cases if c["family"] == f)}
                         for f in args.family},
           "cases": cases}
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")
    print(json.dumps({k: v for k, v in doc.items() if k != "cases"}))


if __name__ == "__main__":
    main()
