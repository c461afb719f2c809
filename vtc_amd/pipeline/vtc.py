#!/usr/bin/env python3
"""VTC inference step, MI355X-native.  CLI-compatible with the reference's `python -m src.pipeline.vtc`
(/root/reference/src/pipeline/vtc.py:422-497): same positional `dataset`, same flags, same outputs

    output/{dataset}/vtc_raw/shard_{id}.parquet      uid, onset, offset, duration, label
    output/{dataset}/vtc_merged/shard_{id}.parquet   same columns after collar merge
    output/{dataset}/vtc_meta/shard_{id}.parquet     per-file metadata (core/metadata.py)
    output/{dataset}/logits/{uid}-logits_dict_t.pt   [--save_logits]

and the same resume rule (a uid present in ANY vtc_meta/shard_*.parquet is skipped, vtc.py:158-173).

What differs is where the work runs: audio decode and the H2D copy run on reader threads while the GPU works, the
whole forward runs in libvtc_amd.so, and threshold -> run-length decoding + the sigmoid summary happen on
the GPU before anything is copied back (the reference does `.cpu()` first, vtc.py:252-281).

Usage:
    python -m vtc_amd.pipeline.vtc chunks30 --checkpoint VTC-2.0/model/best.ckpt
    python -m vtc_amd.pipeline.vtc chunks30 --array_id 3 --array_count 8     # one of 8 GPU workers
"""

from __future__ import annotations

import argparse
import logging
import math
import os
import sys
import threading
import time
from collections import deque
from concurrent.futures import Future
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as pq
import torch

from ..audio import read_pcm, read_plain_wav_into
from ..core.intervals import intervals_to_columns
from ..core.metadata import vtc_error_row, vtc_meta_row_columns
from ..engine import RF_SIZE
from ..packing import OpenRecording, forward_cost_per_window, pack_forward, packed_decode_input, tile_friendly_batch
from ..segma_compat.config import load_config
from ..segma_compat.config.base import Config
from ..segma_compat.inference import logit_bounds
from ..segma_compat.models import Models
from ..segma_compat.utils.encoders import MultiLabelEncoder
from ..utils import (SEGMENT_SCHEMA, add_sample_argument, atomic_write_parquet, get_dataset_paths, hhmmss,
                     balanced_shards, load_completed_ids, load_manifest, log_benchmark, manifest_paths, merge_segments, meta_table,
                     sample_manifest, segments_table_from_blocks, set_seeds, shard_list)

logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s", datefmt="%H:%M:%S", stream=sys.stdout)
logger = logging.getLogger("vtc")

_TLS = threading.local()
# summed wall time per stage (logged at the end).  read = read_file + read_stage (pin + H2D + front-end enqueue) + read_sync
# summed over the reader threads; gpu_run = gpu_launch + gpu_collect + the decode launch; write = parquet flushes
STAGE_SECONDS = {"read": 0.0, "read_file": 0.0, "read_stage": 0.0, "read_sync": 0.0, "gpu_wait": 0.0, "gpu_run": 0.0,
                 "gpu_launch": 0.0, "gpu_collect": 0.0, "finish": 0.0, "write": 0.0}
_TIMER_LOCK = threading.Lock()
PACK_STATS = {"forwards": 0, "windows": 0}     # packed forwards of the last run and the windows they held (main, gpu_worker)


def _tick(name: str, t0: float) -> None:
    with _TIMER_LOCK:
        STAGE_SECONDS[name] += time.perf_counter() - t0

# Windows of 4 s per forward by device memory: the reference's table (src/pipeline/resources.py:180-188), measured there on
# H100 / A40.  On MI355X (288 GB) the hot path with inputs resident saturates once a forward holds ~25,000 frames (the
# persistent GEMMs then run >= 2 full rounds of tiles): at batch 128 x 4 s or 32 x 10 s the rate is within 4 % of the best
# point, and engine memory is 3-12 GB of the card's 288 (tools/batch_sweep.py; the figures of the current kernels are in the
# newest profiles/*_batch_sweep.json and in DESIGN.md section 8, not repeated here).  The top row of the reference's table
# (512 windows of 4 s = 102 k frames) is also the best point here.  What the table leaves out is the window length: the batch
# is scaled so that batch x chunk seconds stays what the table means (512 x 4 s), which gives 208 windows at 10 s.
_BATCH_TABLE = [(80, 512), (48, 256), (32, 192), (24, 128), (16, 64), (0, 32)]


def recommend_vtc_batch_size(vram_gb: float, chunk_duration_s: float = 4.0) -> int:
    """Windows per forward for a card with vram_gb of memory and windows of chunk_duration_s seconds
    (reference: src/pipeline/resources.py:190-196, which knows 4 s windows only)."""
    bs = 32
    for min_vram, b in _BATCH_TABLE:
        if vram_gb >= min_vram:
            bs = b
            break
    if abs(chunk_duration_s - 4.0) > 1e-9:
        bs = max(16, int(round(bs * 4.0 / chunk_duration_s / 16.0)) * 16)
    return bs


def _drop_uids(table: pa.Table, uids: set) -> pa.Table:
    if table.num_rows == 0 or not uids:
        return table
    mask = pc.invert(pc.is_in(table.column("uid"), value_set=pa.array(sorted(uids), pa.string())))
    return table.filter(mask)


def _unique_rows(table: pa.Table) -> pa.Table:
    """Drop exact duplicate rows (reference vtc.py:362-365, `seg_df.unique()`), keeping first occurrences in their
    original order, so the file is deterministic: manifest order, label-major and time-ordered inside a recording."""
    if table.num_rows == 0:
        return table
    names = table.column_names
    first = table.append_column("__row", pa.array(np.arange(table.num_rows, dtype=np.int64))) \
        .group_by(names, use_threads=False).aggregate([("__row", "min")]).column("__row_min").to_numpy()
    if len(first) == table.num_rows:
        return table
    return table.take(pa.array(np.sort(first)))


def _unique_by_uid_keep_last(table: pa.Table) -> pa.Table:
    uids = table.column("uid").to_pylist()
    last = {}
    for i, u in enumerate(uids):
        last[u] = i
    if len(last) == len(uids):
        return table
    return table.take(pa.array(sorted(last.values())))


def stage_audio(path, engine) -> torch.Tensor:
    """Reader-thread stage: read the file's raw PCM (int16 / float32, any channel count and rate), copy it to HBM on this
    thread's own stream and run the GPU front-end there (channel mix, resample to 16 kHz, float32), so that decode,
    H2D and front-end of file i+1 overlap the forward of file i.  Returns only after the waveform is ready."""
    t0 = time.perf_counter()
    pcm, sr = read_pcm(path)
    _tick("read_file", t0)
    t1 = time.perf_counter()
    device = engine.device
    copy_stream = getattr(_TLS, "stream", None)
    if copy_stream is None:
        copy_stream = _TLS.stream = torch.cuda.Stream(device)
    if pcm.nbytes >= 2 * STAGE_CHUNK_BYTES and pcm.flags.c_contiguous:
        # long recording (a day-long file is 1.8 GB of PCM16): through two pinned 64 MB buffers of this thread, so that the
        # page-cache -> pinned copy of piece k+1 runs while piece k crosses PCIe, and pinned memory per reader thread is
        # 128 MB whatever the file size (pinning a file-sized buffer cost as much as reading it)
        bufs = getattr(_TLS, "chunk_bufs", None)
        if bufs is None:
            bufs = _TLS.chunk_bufs = [torch.empty(STAGE_CHUNK_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        flat = pcm.reshape(-1)
        tdt = _TORCH_PCM[pcm.dtype.type]
        per = STAGE_CHUNK_BYTES // flat.dtype.itemsize
        done = [None, None]
        with torch.cuda.stream(copy_stream):
            dev = torch.empty(flat.shape[0], dtype=tdt, device=device)
            for k, a in enumerate(range(0, flat.shape[0], per)):
                j, m = k & 1, min(per, flat.shape[0] - a)
                if done[j] is not None:
                    done[j].synchronize()              # the copy that last used this buffer has left it
                view = bufs[j][: m * flat.dtype.itemsize].view(tdt)
                np.copyto(view.numpy(), flat[a:a + m])
                dev[a:a + m].copy_(view, non_blocking=True)
                done[j] = torch.cuda.Event()
                done[j].record(copy_stream)
            wav = engine.prepare_audio(dev.view(pcm.shape), sr)
    else:
        if isinstance(pcm, np.memmap) or pcm.nbytes >= (8 << 20) or not pcm.flags.writeable:
            host = torch.empty(pcm.shape, dtype=_TORCH_PCM[pcm.dtype.type], pin_memory=True)
            np.copyto(host.numpy(), pcm)                  # page cache / mmap -> pinned staging, no conversion on the host
        else:
            host = torch.from_numpy(pcm)                  # small file, already in memory: a pageable copy is cheaper than pinning
        with torch.cuda.stream(copy_stream):
            wav = engine.prepare_audio(host.to(device, non_blocking=True), sr)
    _tick("read_stage", t1)
    t2 = time.perf_counter()
    copy_stream.synchronize()
    _tick("read_sync", t2)
    _tick("read", t0)
    if wav.numel() < RF_SIZE:
        raise ValueError(f"audio too short: {wav.numel()} samples < receptive field {RF_SIZE}")
    return wav


_TORCH_PCM = {np.int16: torch.int16, np.float32: torch.float32, np.uint8: torch.uint8}   # uint8 = packed 24-bit PCM (n, ch, 3)
STAGE_CHUNK_BYTES = 64 << 20       # long recordings cross PCIe in pieces of this size (two pinned buffers per reader thread)
BUNDLE_FILE_BYTES = 4 << 20        # files up to this size (~2 min of 16 kHz PCM16) are staged in bundles
BUNDLE_BYTES = 16 << 20            # pinned staging buffer of a reader thread (int16 samples of one bundle)
BUNDLE_FILES = 256
PACK_BATCHES = 8                   # recordings of fewer than this many batches of windows share forwards window by window


def plan_reader_tasks(file_bytes: list, step: int, batch_size: int, bundle: bool = True) -> list:
    """Reader tasks in manifest order: each task is a list of consecutive file indices.  A run of short files (<= BUNDLE_FILE_BYTES
    each) becomes ONE task of up to one batch of windows (estimated from the size as if mono 16 kHz PCM16), BUNDLE_BYTES and
    BUNDLE_FILES; a file that is long, of unknown size (0: missing), or alone in its run is a task of its own."""
    tasks, i, n = [], 0, len(file_bytes)
    while i < n:
        first, nbytes, windows = i, 0, 0
        while (bundle and i < n and i - first < BUNDLE_FILES and 0 < file_bytes[i] <= BUNDLE_FILE_BYTES
               and nbytes + file_bytes[i] <= BUNDLE_BYTES):
            w_est = max(1, -(-(file_bytes[i] // 2 - RF_SIZE + 1) // step))
            if i > first and windows + w_est > batch_size:
                break
            windows += w_est
            nbytes += file_bytes[i]
            i += 1
        if i - first >= 2:
            tasks.append(list(range(first, i)))
        else:
            i = first + 1
            tasks.append([first])
    return tasks


def stage_bundle(paths, engine) -> list:
    """Reader-thread stage for a run of short files (thousands of 1-30 s recordings: the per-file cost of stage_audio --
    a pinned allocation, an H2D copy, a front-end launch and a stream synchronisation each -- is then what bounds the run,
    not the GPU).  Mono 16 kHz PCM16 WAVs are read back to back into this thread's pinned buffer and cross PCIe in ONE
    copy with ONE front-end launch (int16 -> float32 is per sample, so the concatenation converts as its parts would);
    the waveforms returned are views of that one device array.  Any other file in the run (another container, rate or
    channel count) goes through stage_audio on its own.  Returns one waveform or one exception per path, in order."""
    t0 = time.perf_counter()
    host = getattr(_TLS, "bundle_host", None)
    if host is None:
        host = _TLS.bundle_host = torch.empty(BUNDLE_BYTES // 2, dtype=torch.int16, pin_memory=True)
    dst = host.numpy()
    out, spans, pos = [None] * len(paths), [], 0
    for k, path in enumerate(paths):
        try:
            n = read_plain_wav_into(path, dst, pos)
        except Exception as e:
            out[k] = e
            continue
        if n is None:
            continue                      # not a plain file: staged on its own below
        spans.append((k, pos, n))
        pos += n
    _tick("read_file", t0)
    t1 = time.perf_counter()
    if pos:
        copy_stream = getattr(_TLS, "stream", None)
        if copy_stream is None:
            copy_stream = _TLS.stream = torch.cuda.Stream(engine.device)
        with torch.cuda.stream(copy_stream):
            arena = engine.prepare_audio(host[:pos].to(engine.device, non_blocking=True), 16000)
        _tick("read_stage", t1)
        t2 = time.perf_counter()
        copy_stream.synchronize()         # the pinned buffer is free again and the waveforms are ready
        _tick("read_sync", t2)
        for k, a, n in spans:
            out[k] = (arena[a:a + n] if n >= RF_SIZE else
                      ValueError(f"audio too short: {n} samples < receptive field {RF_SIZE}"))
    _tick("read", t0)
    for k, path in enumerate(paths):
        if out[k] is None:
            try:
                out[k] = stage_audio(path, engine)
            except Exception as e:
                out[k] = e
    return out


def gpu_stage(engine, gpu_wav: torch.Tensor, lo, hi, batch_size: int, want_logits: bool):
    """GPU-thread stage: forward + threshold/run-length decode; only the decoded triples (and, on request, the
    logits) come back to the host."""
    return gpu_group_stage(engine, [gpu_wav], lo, hi, batch_size, want_logits, grouped=False)[0]


def gpu_group_stage(engine, wavs, lo, hi, batch_size: int, want_logits: bool, grouped: bool = True):
    """GPU-thread stage for several short recordings: their windows share forward passes (vtc_infer_waveforms) and one
    segmented decode with per-recording sigmoid summaries.  Returns one gpu_stage-shaped tuple per recording.
    A 30 s file is 8 windows of 4 s: alone it would use 1/16 of a 128-window batch."""
    forward = gpu_launch(engine, wavs, batch_size, grouped)
    return gpu_collect(engine, gpu_decode_launch(engine, forward, lo, hi, want_logits), want_logits)


def gpu_launch(engine, wavs, batch_size: int, grouped: bool):
    """GPU stage, step 1: enqueue the forward passes on the current stream; nothing waits."""
    cur = torch.cuda.current_stream(engine.device)
    for w in wavs:
        w.record_stream(cur)
    with torch.no_grad():
        if grouped:
            logits, rows = engine.infer_waveforms(wavs, batch_size)
        else:
            logits = engine.infer_waveform(wavs[0], batch_size)
            rows = np.array([0, int(logits.shape[0])], dtype=np.int64)
    return logits, rows, [int(w.numel()) for w in wavs]


def gpu_decode_launch(engine, forward, lo, hi, want_logits: bool):
    """Step 2: enqueue, behind the forward and still without waiting, the threshold / run-length decode
    (vtc_decode_launch: kernels + copies of the results into pinned memory) and, on request, the copy of the logits.
    One decode can be in flight per engine, so the GPU thread's order is: forward(i+1), collect(i), decode_launch(i+1);
    the next forward is then enqueued behind this decode, and the GPU never waits for Python between batches."""
    logits, rows, n_samples = forward
    with torch.no_grad():
        engine.decode_launch(logits, rows, lo, hi, summary=True)
        cpu_logits, copied = None, None
        if want_logits:
            cpu_logits = torch.empty(logits.shape, dtype=logits.dtype, pin_memory=True)
            cpu_logits.copy_(logits, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(torch.cuda.current_stream(engine.device))
    return logits, rows, cpu_logits, copied, n_samples


def gpu_collect(engine, launched, want_logits: bool):
    """Step 3: wait for the decode enqueued by gpu_decode_launch (a forward enqueued after it keeps running) and split
    the runs by recording.  One gpu_stage-shaped tuple per recording."""
    logits, rows, cpu_logits, copied, n_samples = launched
    fidx, starts, ends, lab, mx, mean = engine.decode_collect()
    if copied is not None:
        copied.synchronize()
    bounds = np.searchsorted(fidx, np.arange(len(n_samples) + 1))      # fidx is sorted (recording-major output)
    out = []
    for f, n in enumerate(n_samples):
        a, b = int(bounds[f]), int(bounds[f + 1])
        part = cpu_logits[int(rows[f]):int(rows[f + 1])].clone() if want_logits else None
        out.append((starts[a:b], ends[a:b], lab[a:b], float(mx[f]), float(mean[f]), n, part))
    return out


class NonFiniteLogits(ValueError):
    """The recording's logits hold NaN / inf: an activation left the range of the 16-bit storage type."""


def finish_file(decoded, uid: str, threshold: float, labels, save_to: Path | None):
    """Host stage: decoded runs -> the file's segment columns + meta row (reference vtc.py:252-281).  Columns, not one
    dict per segment: a block is (uid, onset, offset, duration, label_codes) with numpy arrays, joined into the shard's
    table by utils.segments_table_from_blocks; same values as core.intervals.intervals_to_segments / vtc_meta_row."""
    starts, ends, lab, max_logit, mean_sig, _, cpu_logits = decoded
    if not math.isfinite(max_logit) and not max_logit == -math.inf:
        # vtc_decode_collect marks a recording with NaN / inf logits by max_logit = NaN (include/vtc_amd.h): fp16 storage
        # overflowed somewhere upstream; such a file gets an error row, never an empty-looking valid one
        raise NonFiniteLogits("non-finite logits: an activation left the range of the 16-bit storage type (try engine.mfma_dtype: bf16)")
    if save_to is not None:
        torch.save({labels[j]: cpu_logits[:, j] for j in range(len(labels))}, save_to / f"{uid}-logits_dict_t.pt")
    onset, offset, duration, keep = intervals_to_columns(starts, ends)
    codes = np.asarray(lab, dtype=np.int32)
    if not keep.all():
        onset, offset, duration, codes = onset[keep], offset[keep], duration[keep], codes[keep]
    max_sig = round(float(torch.sigmoid(torch.tensor(max_logit, dtype=torch.float32)).item()), 4)
    row = vtc_meta_row_columns(uid, threshold, duration, codes, labels, max_sig, round(float(mean_sig), 4))
    return (uid, onset, offset, duration, codes), row


def block_rows(block, labels) -> list[dict]:
    """A segment block as the reference's list of dicts (tests, small tools)."""
    uid, onset, offset, duration, codes = block
    return [{"uid": uid, "onset": float(a), "offset": float(b), "duration": float(d), "label": labels[int(k)]}
            for a, b, d, k in zip(onset, offset, duration, codes)]


def process_file(engine, wav: np.ndarray, uid: str, threshold: float, labels, batch_size: int, save_to: Path | None):
    """One in-memory waveform through the fused GPU path -> (segment rows, meta row), synchronously."""
    if wav.shape[0] < RF_SIZE:
        raise ValueError(f"audio too short: {wav.shape[0]} samples < receptive field {RF_SIZE}")
    lo, hi = logit_bounds({l: {"lower_bound": threshold, "upper_bound": 1.0} for l in labels}, labels)
    decoded = gpu_stage(engine, torch.from_numpy(wav).to(engine.device), lo, hi, batch_size, save_to is not None)
    return finish_file(decoded, uid, threshold, labels, save_to)


def main(dataset: str, config: str = "VTC-2.0/model/config.yml", checkpoint: str = "VTC-2.0/model/best.ckpt",
         threshold: float = 0.5, min_duration_on_s: float = 0.1, min_duration_off_s: float = 0.3, batch_size: int = 0,
         save_logits: bool = False, device: str = "cuda", array_id: int | None = None,
         array_count: int | None = None, sample=None, prefetch: int = 4, balance: bool = False,
         flush_every: int = 0, flush_seconds: float = 300.0, no_group: bool = False):
    set_seeds(42)
    # Reader threads, the GPU thread and this thread all run Python between their GIL-free calls (file reads, copies, the
    # library).  With CPython's default 5 ms switch interval a thread that comes back from such a call waits up to 5 ms for the
    # GIL: at 3,000 short files per second that convoy was a third of the GPU thread's time.
    sys.setswitchinterval(float(os.environ.get("VTC_SWITCH_INTERVAL", "0.0005")))
    pin = os.environ.get("VTC_WORKER_CPUS", "")
    if pin:      # the launcher's CPU slice for this worker: before any thread pool starts, so every thread inherits it
        try:
            os.sched_setaffinity(0, {int(c) for c in pin.split(",") if c.strip()})
        except (OSError, ValueError) as e:
            logger.warning(f"could not pin to CPUs {pin}: {e}")
    for k in STAGE_SECONDS:
        STAGE_SECONDS[k] = 0.0
    if device != "cuda":
        raise SystemExit(f"vtc_amd runs on ROCm GPUs only (--device {device} requested); it has no CPU path")
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm GPU visible; vtc_amd has no CPU path")
    dev_index = torch.cuda.current_device()
    if Path(config).exists():
        model_config: Config = load_config(config)
    else:
        logger.warning(f"config {config} not found; using built-in defaults (4 s windows, VTC_LABELS order)")
        model_config = Config()
    if batch_size <= 0:
        props = torch.cuda.get_device_properties(dev_index)
        ladder = recommend_vtc_batch_size(props.total_memory // (1024 ** 3), float(model_config.audio.chunk_duration_s))
        batch_size = tile_friendly_batch(ladder, float(model_config.audio.chunk_duration_s))
        logger.info(f"Auto batch_size={batch_size} for {props.name}, {model_config.audio.chunk_duration_s:g} s windows "
                    f"(memory ladder {ladder}, trimmed to whole rounds of GEMM tiles)")
    else:
        # an explicit batch is respected; say so when a nearby one would fill the GEMMs' last round of tiles much better
        chunk = float(model_config.audio.chunk_duration_s)
        near = tile_friendly_batch(int(batch_size * 1.3), chunk)
        mine, best = forward_cost_per_window(batch_size, chunk), forward_cost_per_window(near, chunk)
        if mine and best and mine / best > 1.03:
            logger.info(f"batch_size {batch_size}: its GEMM launches leave part of their last round of tiles empty; "
                        f"{near} windows per forward would be ~{100 * (mine / best - 1):.0f} % faster per window")

    paths = get_dataset_paths(dataset)
    logger.info(f"Dataset: {dataset}")
    logger.info(f"  manifest  : {paths.manifest}")
    logger.info(f"  output    : {paths.output}")
    logger.info(f"  threshold : {threshold}")
    logger.info(f"  batch_size: {batch_size}")

    manifest = sample_manifest(load_manifest(paths.manifest), sample)
    resolved = manifest_paths(manifest)
    file_ids = [Path(p).stem for p in resolved]
    uid_to_path = dict(zip(file_ids, resolved))
    if array_id is not None and array_count is not None:
        if balance:
            # duration-balanced shards for skewed manifests (every worker derives the same assignment)
            costs = []
            for p in resolved:
                try:
                    costs.append(os.path.getsize(p))
                except OSError:
                    costs.append(0)
            file_ids = balanced_shards(file_ids, costs, array_count)[array_id]
        else:
            file_ids = shard_list(file_ids, array_id, array_count)
        logger.info(f"Shard {array_id}/{array_count - 1}: {len(file_ids)} files")
    shard_id = array_id if array_id is not None else 0

    # ---- resume: skip files already processed by ANY shard
    meta_dir = paths.output / "vtc_meta"
    meta_path = meta_dir / f"shard_{shard_id}.parquet"
    completed = load_completed_ids(meta_dir, id_column="uid", pattern="shard_*.parquet")
    prev_meta = pq.read_table(meta_path) if meta_path.exists() else None
    todo = [u for u in file_ids if u not in completed]
    if len(todo) < len(file_ids):
        logger.info(f"Resume: {len(file_ids) - len(todo)} done, {len(todo)} remaining")
    if not todo and not file_ids:
        logger.info("No files to process.")
        return

    # ---- model
    l_encoder = MultiLabelEncoder(labels=model_config.data.classes)
    model = Models[model_config.model.name].load_from_checkpoint(
        checkpoint_path=checkpoint, label_encoder=l_encoder, config=model_config, train=False)
    model.eval()
    model.to(torch.device("cuda", dev_index))
    chunk_duration_s = model_config.audio.chunk_duration_s
    engine = model.engine(chunk_duration_s, batch_size)
    labels = l_encoder.labels

    logits_dir = paths.output / "logits"
    if save_logits:
        logits_dir.mkdir(parents=True, exist_ok=True)

    sizes = {}
    for u in todo:
        try:
            sizes[u] = Path(uid_to_path[u]).stat().st_size
        except OSError:
            sizes[u] = 0
    total_bytes = sum(sizes.values()) or 1
    total = len(todo)
    print(f"Shard {shard_id}: {total} files, {total_bytes / 1e9:.1f} GB", flush=True)

    meta_rows, seg_blocks = [], []
    n_errors, bytes_done, audio_seconds = 0, 0, 0.0
    t0 = last_log = last_flush = time.time()
    since_flush = 0
    log_every = max(1, total // 20)
    raw_path = paths.output / "vtc_raw" / f"shard_{shard_id}.parquet"
    prev_seg_all = pq.read_table(raw_path).cast(SEGMENT_SCHEMA) if (raw_path.exists() and completed) else None

    def write_outputs(final: bool):
        """Shard outputs from the rows gathered so far, merged with the shard's previous rows (resume) and
        de-duplicated; tmp + rename, so readers and a later resume never see a torn file.  Called at the end and,
        for durable progress, every --flush_every files / --flush_seconds seconds (reference writes once, vtc.py:375-387)."""
        new_meta = meta_table(meta_rows) if meta_rows else None
        parts = []
        if prev_meta is not None:
            kept = _drop_uids(prev_meta, {r["uid"] for r in meta_rows}) if new_meta is not None else prev_meta
            if kept.num_rows:
                parts.append(kept.cast(new_meta.schema) if new_meta is not None else kept)
        if new_meta is not None:
            parts.append(new_meta)
        meta_df = pa.concat_tables(parts) if parts else meta_table([])
        before = meta_df.num_rows
        meta_df = _unique_by_uid_keep_last(meta_df)
        if final and meta_df.num_rows < before:
            logger.warning(f"Dedup: removed {before - meta_df.num_rows} duplicate meta rows (shard {shard_id})")

        new_seg = segments_table_from_blocks(seg_blocks, labels)
        if prev_seg_all is not None:
            prev_seg = _drop_uids(prev_seg_all, {b[0] for b in seg_blocks})
            seg_df = pa.concat_tables([t for t in (prev_seg, new_seg) if t.num_rows]) if (prev_seg.num_rows or new_seg.num_rows) else new_seg
        else:
            seg_df = new_seg
        before = seg_df.num_rows
        seg_df = _unique_rows(seg_df)
        if final and seg_df.num_rows < before:
            logger.warning(f"Dedup: removed {before - seg_df.num_rows} duplicate segment rows (shard {shard_id})")

        # segments first, meta last: a uid only counts as done (resume) once its segments are on disk
        atomic_write_parquet(seg_df, raw_path)
        merged_df = merge_segments(seg_df, min_duration_off_s, min_duration_on_s)
        atomic_write_parquet(merged_df, paths.output / "vtc_merged" / f"shard_{shard_id}.parquet")
        atomic_write_parquet(meta_df, meta_path)
        return seg_df, merged_df

    # Three overlapped stages: reader threads (decode + H2D on their own streams) -> one GPU thread (forward +
    # decode, in manifest order) -> this thread (segment rows, meta rows).  Per-file failures at any stage
    # become error rows and the run continues (reference vtc.py:245-249).
    lo, hi = logit_bounds({l: {"lower_bound": threshold, "upper_bound": 1.0} for l in labels}, labels)
    depth = max(1, prefetch)
    save_to = logits_dir if save_logits else None

    step = engine.step
    results = [Future() for _ in range(total)]
    stop = threading.Event()
    # bytes of files being staged ahead of the GPU (short files are read far ahead to fill batches; two day-long recordings
    # of 1.8 GB fit, so the next one is read while the GPU is still on the one before the current).  Host cost: pinned
    # memory is bounded per reader thread (128 MB + 16 MB), not by this; device cost: ~2-3 x this in float32 waveforms.
    read_budget = int(float(os.environ.get("VTC_READ_BUDGET_GB", "4")) * (1 << 30))
    max_ahead = 256
    bundle = not no_group and os.environ.get("VTC_NO_BUNDLE") != "1"     # A/B switch: stage every file on its own
    pack = not no_group and os.environ.get("VTC_NO_PACK") != "1"         # A/B switch: whole recordings per forward (round 2's grouping)
    PACK_STATS["forwards"] = PACK_STATS["windows"] = 0

    def gpu_worker(readers):
        """Pulls staged recordings in manifest order; recordings shorter than a batch are grouped until the batch is
        full (or until waiting for the next one would idle the GPU) and run together; long ones run alone."""
        torch.cuda.set_device(engine.device)
        inflight, nxt, inflight_bytes = deque(), 0, 0
        group, group_windows = [], 0

        tasks = plan_reader_tasks([sizes.get(u, 0) for u in todo], step, batch_size, bundle)

        def top_up():
            """Submit reader tasks in manifest order (plan_reader_tasks: one recording, or a bundle of consecutive short files,
            so that thousands of 1-30 s files cost one reader task, one copy and one front-end launch per forward)."""
            nonlocal nxt, inflight_bytes
            # two tasks in flight whatever their size (the one the GPU works on and the next one); beyond that only
            # what fits the byte budget, so that a manifest of day-long recordings (3.7 GB of float32 each in HBM) does not
            # stage `prefetch` of them per worker.  `prefetch` sizes the reader pool, not this.
            while nxt < len(tasks):
                idx = tasks[nxt]
                nbytes = sum(sizes.get(todo[i], 0) for i in idx)
                if not (len(inflight) < 2 or (len(inflight) < max_ahead and inflight_bytes + nbytes <= read_budget)):
                    break
                if len(idx) > 1:
                    fut = readers.submit(stage_bundle, [uid_to_path[todo[i]] for i in idx], engine)
                else:
                    fut = readers.submit(stage_audio, uid_to_path[todo[idx[0]]], engine)
                inflight.append((idx, nbytes, fut))
                inflight_bytes += nbytes
                nxt += 1

        pending = None     # (items, launched): forward enqueued, decode not collected yet
        serial_gpu = os.environ.get("VTC_SERIAL_GPU_STAGE") == "1"    # diagnostic: collect right after launching (A/B timing)
        carry, carry_windows = deque(), 0      # window packing: recordings with windows still to run (pack_forward)

        def drain():
            """Collect the batch whose forward + decode are in flight (host sync) and publish its results."""
            nonlocal pending
            if pending is None:
                return
            (items, launched), pending = pending, None
            t_col = time.perf_counter()
            try:
                outs = gpu_collect(engine, launched, save_logits)
                for (i, _), o in zip(items, outs):
                    results[i].set_result(o)
            except Exception as e:
                redo(items, e)
            _tick("gpu_collect", t_col)

        def redo(items, e):
            engine.decode_abandon()      # a decode left in flight by the failed step must not block the retries
            if len(items) == 1:
                results[items[0][0]].set_exception(e)
                return
            for i, w in items:       # one bad recording must not take its batch mates down: redo them one by one
                try:
                    results[i].set_result(gpu_stage(engine, w, lo, hi, batch_size, save_logits))
                except Exception as e1:
                    results[i].set_exception(e1)

        def redo_alone(victims, e):
            """A packed forward (or its decode) failed: every recording it touched runs again on its own."""
            logger.warning(f"packed forward failed ({e}); {len(victims)} recordings run one by one")
            engine.decode_abandon()
            for i, w in victims:
                try:
                    results[i].set_result(gpu_stage(engine, w, lo, hi, batch_size, save_logits))
                except Exception as e1:
                    results[i].set_exception(e1)

        def pump(flush: bool):
            """Window packing: run full forwards while the queued recordings hold a batch of windows (with `flush`, also the
            partial one that is left), each in the order forward(k) -> collect(k-1) -> decode_launch(recordings that forward k
            completed)."""
            nonlocal carry_windows, pending
            while carry_windows >= batch_size or (flush and carry_windows):
                nw = min(batch_size, carry_windows)
                touched = [(o.i, o.wav) for o in carry]
                t_run = time.perf_counter()
                try:
                    done, out = pack_forward(engine, carry, nw)
                except Exception as e:
                    carry.clear()
                    carry_windows = 0
                    drain()
                    redo_alone(touched, e)
                    return
                carry_windows -= nw
                PACK_STATS["forwards"] += 1
                PACK_STATS["windows"] += nw
                _tick("gpu_launch", t_run)
                drain()
                if done:
                    items = [(o.i, o.wav) for o in done]
                    try:
                        pending = (items, gpu_decode_launch(engine, packed_decode_input(done, out), lo, hi, save_logits))
                    except Exception as e:
                        redo_alone(items, e)
                    if serial_gpu:
                        drain()
                _tick("gpu_run", t_run)

        def run_group():
            """Enqueue the forward of the current group, collect the previous group (its forward and decode were
            enqueued one call ago), then enqueue this group's decode behind its forward."""
            nonlocal group, group_windows, pending
            if not group:
                return
            items, group, group_windows = group, [], 0
            t_run = time.perf_counter()
            try:
                forward = gpu_launch(engine, [w for _, w in items], batch_size, len(items) > 1)
            except Exception as e:
                drain()
                redo(items, e)
                return
            _tick("gpu_launch", t_run)
            drain()
            try:
                pending = (items, gpu_decode_launch(engine, forward, lo, hi, save_logits))
            except Exception as e:
                redo(items, e)
                return
            if serial_gpu:
                drain()
            _tick("gpu_run", t_run)

        try:
            top_up()
            while inflight and not stop.is_set():
                idx, nbytes, fut = inflight.popleft()
                inflight_bytes -= nbytes
                top_up()
                if not fut.done():
                    run_group()          # the next recording is not staged yet: do not hold a partial batch back
                    pump(True)
                    t_run = time.perf_counter()
                    drain()              # ... and publish what is in flight instead of sitting on it while waiting
                    _tick("gpu_run", t_run)
                t_wait = time.perf_counter()
                try:
                    staged = fut.result()
                    staged = staged if len(idx) > 1 else [staged]
                except Exception as e:
                    staged = [e] * len(idx)
                finally:
                    _tick("gpu_wait", t_wait)
                for i, wav in zip(idx, staged):
                    if isinstance(wav, BaseException):
                        results[i].set_exception(wav)
                        continue
                    n_win = -(-(int(wav.numel()) - RF_SIZE + 1) // step)
                    if pack and n_win < PACK_BATCHES * batch_size:
                        carry.append(OpenRecording(i, wav, n_win))
                        carry_windows += n_win
                        pump(False)
                        continue
                    if no_group or n_win >= batch_size:
                        pump(True)       # manifest order: what is queued runs before the long recording
                        run_group()
                        group, group_windows = [(i, wav)], n_win
                        run_group()
                        continue
                    if group and group_windows + n_win > batch_size:
                        run_group()
                    group.append((i, wav))
                    group_windows += n_win
                    if group_windows >= batch_size:
                        run_group()
            run_group()
            pump(True)
            drain()
        except BaseException as e:       # never leave the consumer waiting on a future that will not resolve
            for f in results:
                if not f.done():
                    f.set_exception(RuntimeError(f"GPU worker stopped: {e}"))
            raise
        finally:
            for _, _, f in inflight:
                f.cancel()
            for f in results:
                if not f.done():
                    f.set_exception(RuntimeError("run stopped before this file was processed"))

    # A recording whose activations leave the fp16 range (65,504) comes back with NaN logits.  Rather than lose the file,
    # it is run again, alone, by a second engine on the same weights with bf16 storage (same kernels, 8 bits of mantissa:
    # DESIGN.md section 3 gives what that costs in accuracy).  Off with engine.bf16_rescue: false; pointless when the
    # configured type is bf16 already.
    can_rescue = bool(getattr(model_config.engine, "bf16_rescue", True)) and model_config.engine.mfma_dtype == "f16"
    n_rescued = 0

    def rescue(uid: str):
        eng = model.rescue_engine(chunk_duration_s, min(batch_size, 16))
        wav = stage_audio(uid_to_path[uid], eng)
        decoded = gpu_stage(eng, wav, lo, hi, min(batch_size, 16), save_logits)
        return decoded, finish_file(decoded, uid, threshold, labels, save_to)

    with ThreadPoolExecutor(max_workers=depth) as readers, ThreadPoolExecutor(max_workers=1) as gpu_thread:
        worker = gpu_thread.submit(gpu_worker, readers)
        try:
            for i, uid in enumerate(todo):
                try:
                    decoded = results[i].result()
                    t_fin = time.perf_counter()
                    segs, row = finish_file(decoded, uid, threshold, labels, save_to)
                    _tick("finish", t_fin)
                    audio_seconds += decoded[5] / 16000.0
                except Exception as e:  # per-file failure -> error row, run continues (reference vtc.py:245-249)
                    rescued = None
                    if isinstance(e, NonFiniteLogits) and can_rescue:
                        try:
                            decoded, rescued = rescue(uid)
                            audio_seconds += decoded[5] / 16000.0
                            n_rescued += 1
                            logger.warning(f"{uid}: an activation left the fp16 range; the file was run again with bf16 storage")
                        except Exception as e2:
                            e = e2
                    if rescued is None:
                        n_errors += 1
                        logger.warning(f"{uid}: {e}")
                        meta_rows.append(vtc_error_row(uid, str(e)))
                        continue
                    segs, row = rescued
                seg_blocks.append(segs)
                meta_rows.append(row)
                bytes_done += sizes.get(uid, 0)
                now = time.time()
                since_flush += 1
                if i + 1 < total and ((flush_every > 0 and since_flush >= flush_every) or
                                      (flush_seconds > 0 and now - last_flush >= flush_seconds)):
                    t_w = time.perf_counter()
                    write_outputs(final=False)
                    _tick("write", t_w)
                    since_flush, last_flush = 0, time.time()
                if (i + 1) % log_every == 0 or i + 1 == total or now - last_log >= 60:
                    rate = bytes_done / max(now - t0, 1e-9)
                    eta_s = (total_bytes - bytes_done) / rate if rate > 0 else 0
                    eta = f"{eta_s / 60:.0f}m" if eta_s < 3600 else f"{eta_s / 3600:.1f}h"
                    print(f"  VTC  {i + 1:>4}/{total}  {bytes_done / 1e9:.1f}/{total_bytes / 1e9:.1f} GB"
                          f" ({100.0 * bytes_done / total_bytes:.0f}%)  ETA {eta}", flush=True)
                    last_log = now
        finally:
            stop.set()          # an interrupt here must not leave the worker chewing through the rest of the manifest
            model.close_rescue()
        worker.result()

    t_w = time.perf_counter()
    seg_df, merged_df = write_outputs(final=True)
    _tick("write", t_w)

    t_end = time.time()
    wall = t_end - t0
    print(flush=True)
    logger.info("-" * 50)
    logger.info(f"Shard {shard_id} complete")
    logger.info(f"  Files     : {total - n_errors}/{total}  ({n_errors} errors" + (f", {n_rescued} re-run in bf16)" if n_rescued else ")"))
    logger.info(f"  Threshold : {threshold}")
    logger.info(f"  Segments  : {seg_df.num_rows:,} raw, {merged_df.num_rows:,} merged")
    logger.info(f"  Wall time : {hhmmss(wall)}   ({audio_seconds / 3600.0 / max(wall, 1e-9):.3f} audio-hours/s)")
    logger.info("  Stage time: " + ", ".join(f"{k} {v:.2f}s" for k, v in STAGE_SECONDS.items()) +
                f"  (read is summed over {depth} reader threads, gpu_run = gpu_launch + gpu_collect; wall {wall:.2f}s)")
    logger.info("-" * 50)
    done_bytes = sum(os.path.getsize(uid_to_path[u]) for u in todo if os.path.exists(uid_to_path[u]))
    log_benchmark(step="vtc", dataset=dataset, n_files=total, wall_seconds=wall, total_audio_seconds=audio_seconds,
                  total_bytes=done_bytes, n_workers=1,
                  extra={"device": torch.cuda.get_device_name(dev_index), "shard_id": shard_id,
                         "mfma_dtype": model_config.engine.mfma_dtype, "bf16_rescued_files": n_rescued, "loop_start_unix": round(t0, 3), "loop_end_unix": round(t_end, 3),
                         "packed_forwards": PACK_STATS["forwards"], "packed_windows": PACK_STATS["windows"],
                         "stage_seconds": {k: round(v, 3) for k, v in STAGE_SECONDS.items()}})


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="VTC inference with fixed per-file thresholding (MI355X-native).",
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("dataset", help="Dataset name - used to derive manifests/, output/ and figures/ paths.")
    p.add_argument("--config", default="VTC-2.0/model/config.yml", help="model config YAML")
    p.add_argument("--checkpoint", default="VTC-2.0/model/best.ckpt",
                   help="model checkpoint (.ckpt/.pt/.safetensors/.npz) or 'synthetic[:seed]'")
    p.add_argument("--threshold", type=float, default=0.5, help="Sigmoid threshold (default: 0.5)")
    p.add_argument("--min_duration_on_s", type=float, default=0.1, help="Remove segments shorter than this (0.1 s)")
    p.add_argument("--min_duration_off_s", type=float, default=0.3, help="Merge same-label segments with gap <= this (0.3 s)")
    p.add_argument("--batch_size", type=int, default=0, help="Windows per forward pass; 0 = from GPU memory")
    p.add_argument("--save_logits", action="store_true", help="Save per-file logits to output/{dataset}/logits/")
    p.add_argument("--device", default="cuda", choices=["cuda", "cpu", "mps"], help="Only 'cuda' (ROCm) is supported")
    p.add_argument("--array_id", type=int, help="Shard index (SLURM array task ID or launcher rank)")
    p.add_argument("--array_count", type=int, help="Number of shards")
    p.add_argument("--prefetch", type=int, default=4,
                   help="reader threads (container decode + H2D + GPU front-end of the files ahead of the GPU); how many files "
                        "are staged ahead is bounded by bytes (1 GiB), not by this")
    p.add_argument("--balance", action="store_true",
                   help="shard by file size (longest-processing-time) instead of equal counts; all workers of a run "
                        "must agree on this flag")
    p.add_argument("--flush_every", type=int, default=0,
                   help="also write the shard's parquet files after every N finished files (0 = only by time / at the end)")
    p.add_argument("--flush_seconds", type=float, default=300.0,
                   help="write the shard's parquet files at least this often while running (0 = only at the end); a crash "
                        "then loses at most this much work and the resume rule picks up from the flushed files")
    p.add_argument("--no_group", action="store_true",
                   help="one forward per file even when a file is shorter than a batch of windows (the reference's pattern; A/B timing)")
    add_sample_argument(p)
    return p


if __name__ == "__main__":
    main(**vars(build_parser().parse_args()))
