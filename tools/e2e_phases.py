#!/usr/bin/env python3
"""Per-phase host timing of one file through the CLI's process_file steps (diagnostic for tools/e2e_bench.py)."""
import sys, time, tempfile, os
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vtc_amd import synth
from vtc_amd.audio import read_audio, write_wav
from vtc_amd.engine import VtcEngine
from vtc_amd.core.intervals import intervals_to_segments
from vtc_amd.core.metadata import vtc_meta_row
from vtc_amd.segma_compat.inference import logit_bounds

d = Path(tempfile.mkdtemp(dir="/dev/shm"))
write_wav(d / "a.wav", synth.synth_audio(600.0, seed=1000))
eng = VtcEngine(synth.make_weights(0), n_labels=4, chunk_duration_s=10.0, max_batch=64)
labels = ["FEM", "MAL", "KCHI", "OCH"]
lo, hi = logit_bounds({l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in labels}, labels)
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T.setdefault(name, []).append(time.perf_counter() - t0)
for it in range(6):
    t = time.perf_counter(); wav = read_audio(d / "a.wav"); tick("read_audio", t)
    t = time.perf_counter(); host = torch.from_numpy(wav).pin_memory(); tick("pin", t)
    t = time.perf_counter(); gpu = host.to(eng.device, non_blocking=True); tick("h2d", t)
    t = time.perf_counter(); logits = eng.infer_waveform(gpu, 64); tick("infer", t)
    t = time.perf_counter(); s, e, lab, mx, mean = eng.decode(logits, lo, hi); tick("decode", t)
    t = time.perf_counter(); iv = [(int(a), int(b), labels[int(k)]) for a, b, k in zip(s, e, lab)]; tick("triples", t)
    t = time.perf_counter(); segs = intervals_to_segments(iv, "a"); tick("segments", t)
    t = time.perf_counter(); row = vtc_meta_row("a", 0.5, segs, 0.5, 0.5); tick("meta", t)
for k, v in T.items():
    print(f"{k:12s} {1e3 * float(np.median(v[1:])):8.3f} ms")
print("segments", len(segs))
