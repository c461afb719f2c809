#!/usr/bin/env python3
"""Host budget of the reader stage (no GPU needed): CPU-seconds one reader thread spends per audio-hour, per input
format, measured on this machine's cores, and from that the cores eight workers need to keep 8 GPUs fed.

The reader stage (vtc_amd/pipeline/vtc.py: stage_audio) does, per file: container parse / decode to raw PCM
(vtc_amd.audio.read_pcm; WAV is a memory map, FLAC is decoded by libvtc_amd.so's vtc_flac_decode outside the GIL), then
one copy of the PCM into pinned staging memory.  Channel mix, resampling and float conversion run on the GPU.

    python tools/host_budget.py --seconds 120 --target 15 --out profiles/r02_host_budget.json
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vtc_amd import synth  # noqa: E402
from vtc_amd.audio import read_pcm  # noqa: E402


def cpu_seconds(fn, repeat: int) -> float:
    best = 1e30
    for _ in range(repeat):
        t0 = time.thread_time()
        w0 = time.perf_counter()
        fn()
        best = min(best, max(time.thread_time() - t0, 0.0) if os.environ.get("VTC_BUDGET_THREAD_TIME") else time.perf_counter() - w0)
    return best


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0, help="length of the test recordings")
    ap.add_argument("--target", type=float, default=15.0, help="audio-hours/s one GPU consumes (bench.py value)")
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    os.environ.setdefault("VTC_FLAC_THREADS", "1")          # per-thread cost: the parallel decoder would hide it
    from scipy.io import wavfile
    from oracle import flac_codec                            # a dev tool may use the test encoder to make inputs
    tmp = Path(tempfile.mkdtemp(prefix="vtc_budget_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
    mono = synth.synth_audio(args.seconds, seed=77)
    pcm16 = np.clip(np.round(mono * 32768.0), -32768, 32767).astype(np.int16)
    n44 = int(args.seconds * 44100)
    t = np.arange(n44) / 44100.0
    st44 = np.stack([np.interp(t, np.arange(len(mono)) / 16000.0, mono), np.interp(t, np.arange(len(mono)) / 16000.0, mono[::-1])], axis=1)
    st44_16 = np.clip(np.round(st44 * 32768.0), -32768, 32767).astype(np.int16)
    cases = {}
    wavfile.write(tmp / "mono16k.wav", 16000, pcm16)
    cases["wav_pcm16_16k_mono"] = tmp / "mono16k.wav"
    wavfile.write(tmp / "st44k.wav", 44100, st44_16)
    cases["wav_pcm16_44k1_stereo"] = tmp / "st44k.wav"
    (tmp / "fixed.flac").write_bytes(flac_codec.encode(pcm16.astype(np.int64), 16000, subframe="fixed2"))
    cases["flac_fixed_16k_mono"] = tmp / "fixed.flac"
    (tmp / "lpc.flac").write_bytes(flac_codec.encode(pcm16.astype(np.int64), 16000, subframe="lpc"))
    cases["flac_lpc8_16k_mono"] = tmp / "lpc.flac"
    short = st44_16[: 44100 * min(int(args.seconds), 30)]
    (tmp / "st44k.flac").write_bytes(flac_codec.encode(short.astype(np.int64), 44100, subframe="lpc"))
    cases["flac_lpc8_44k1_stereo"] = tmp / "st44k.flac"
    seconds = {k: args.seconds for k in cases}
    seconds["flac_lpc8_44k1_stereo"] = len(short) / 44100.0

    rows = {}
    for name, path in cases.items():
        staging = {}

        def work():
            pcm, sr = read_pcm(path)
            buf = staging.get("b")
            if buf is None or buf.shape != pcm.shape or buf.dtype != pcm.dtype:
                buf = staging["b"] = np.empty(pcm.shape, pcm.dtype)
            np.copyto(buf, pcm)                               # the copy into (pinned) staging memory
        s = cpu_seconds(work, args.repeat)
        per_hour = s * 3600.0 / seconds[name]
        rows[name] = {"file_bytes": path.stat().st_size, "reader_seconds_per_audio_hour": round(per_hour, 4),
                      "audio_hours_per_s_per_thread": round(1.0 / per_hour, 2),
                      "reader_threads_for_one_gpu": round(args.target * per_hour, 2),
                      "reader_cores_for_node": round(args.gpus * args.target * per_hour, 1)}
        print(name, rows[name], flush=True)
    doc = {"what": "reader stage of vtc_amd.pipeline.vtc (container decode + one copy into staging memory) timed on one thread; "
                   "mix / resample / float conversion are GPU work and not in these figures",
           "machine": {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))},
           "recording_seconds": args.seconds, "target_audio_hours_per_s_per_gpu": args.target, "gpus": args.gpus,
           "plus_per_worker": "1 GPU thread + 1 main thread (segment rows, parquet) per worker: 2 cores",
           "formats": rows}
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")
    print(json.dumps(doc))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
