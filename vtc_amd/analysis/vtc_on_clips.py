#!/usr/bin/env python3
"""VTC on packaged clips, MI355X-native.  CLI-compatible with the reference's
`python -m src.analysis.vtc_on_clips` (/root/reference/src/analysis/vtc_on_clips.py:74-338): for every clip in
output/{dataset}/shards/*.tar (WebDataset layout, members `{clip_id}.{wav|flac}` + side files) it writes

    output/{dataset}/vtc_clips/vtc_raw/shard_{id}.parquet      clip_id, onset, offset, duration, label
    output/{dataset}/vtc_clips/vtc_merged/shard_{id}.parquet   same columns after the collar merge

with onsets/offsets relative to the clip, the same resume rule (a clip_id present in ANY vtc_raw/shard_*.parquet is
skipped, :119-125) and the same per-clip error behaviour (log, count, continue, :204-207).

What differs is where the work runs: clips are decoded from the tar in memory (the reference writes each one to a
temporary file, :177-184), their PCM goes to the GPU front-end, and clips are *batched together*: most clips are a
few windows long, so windows of many clips share one forward pass (vtc_infer_waveforms) and one segmented
threshold/run-length decode (vtc_decode_files) instead of one small launch chain per clip.

Usage:
    python -m vtc_amd.analysis.vtc_on_clips seedlings_10 --checkpoint VTC-2.0/model/best.ckpt
"""

from __future__ import annotations

import argparse
import io
import logging
import queue
import sys
import tarfile
import threading
import os
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as pq
import torch

from ..audio import decode_flac, to_float32
from ..core.intervals import intervals_to_columns
from ..engine import RF_SIZE
from ..packing import OpenRecording, pack_forward, packed_decode_input, tile_friendly_batch
from ..segma_compat.config import load_config
from ..segma_compat.config.base import Config
from ..segma_compat.inference import logit_bounds
from ..segma_compat.models import Models
from ..segma_compat.utils.encoders import MultiLabelEncoder
from ..utils import (atomic_write_parquet, get_dataset_paths, hhmmss, merge_segments, segments_table_from_blocks, set_seeds,
                     shard_list)

logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(message)s", datefmt="%H:%M:%S", stream=sys.stdout)
logger = logging.getLogger("vtc_clips")

CLIP_SCHEMA = pa.schema([("clip_id", pa.string()), ("onset", pa.float64()), ("offset", pa.float64()),
                         ("duration", pa.float64()), ("label", pa.string())])


def iter_tar_samples(tar_path):
    """WebDataset grouping: consecutive members that share the name up to the first dot of the basename form one
    sample {"__key__": key, ext: bytes, ...}."""
    with tarfile.open(tar_path, "r:*") as tf:
        key, sample = None, {}
        for member in tf:
            if not member.isfile():
                continue
            base = member.name
            slash = base.rfind("/")
            dot = base.find(".", slash + 1)
            if dot < 0:
                continue
            k, ext = base[:dot], base[dot + 1:].lower()
            if k != key:
                if key is not None:
                    yield sample
                key, sample = k, {"__key__": k}
            sample[ext] = tf.extractfile(member).read()
        if key is not None:
            yield sample


def decode_clip(sample):
    """Audio member of a sample -> (pcm int16/float32 (n,) or (n, ch), sample_rate).  flac is preferred, wav is the
    fall-back, as in the reference (src/analysis/vtc_on_clips.py:163-169); flac goes through the library's own decoder
    (vtc_flac_decode), straight from the tar member's bytes."""
    if "flac" in sample:
        return decode_flac(sample["flac"])
    if "wav" in sample:
        from scipy.io import wavfile

        sr, data = wavfile.read(io.BytesIO(sample["wav"]))
        if data.dtype not in (np.int16, np.float32):
            data = to_float32(np.asarray(data))
        return np.array(data, order="C"), int(sr)          # own, writable copy (torch.from_numpy)
    raise ValueError("no audio found in sample")


def clips_table(blocks, labels) -> pa.Table:
    """Per-clip column blocks (clip_id, onset, offset, duration, label codes) -> the clip segment table."""
    return segments_table_from_blocks(blocks, labels).rename_columns(CLIP_SCHEMA.names).cast(CLIP_SCHEMA)


def launch_group(engine, group, batch_size):
    """group: list of (clip_id, device waveform).  Enqueue one batched forward over all clips; nothing waits."""
    cur = torch.cuda.current_stream(engine.device)
    for _, w in group:
        w.record_stream(cur)                 # staged on the reader thread's stream
    with torch.no_grad():
        logits, file_rows = engine.infer_waveforms([w for _, w in group], batch_size)
    return logits, file_rows


def collect_group(engine, group):
    """Wait for the decode launched for `group` and turn its runs into one column block per clip (same values as
    core.intervals.intervals_to_segments on every clip)."""
    fidx, starts, ends, lab = engine.decode_collect()
    onset, offset, duration, keep = intervals_to_columns(starts, ends)
    bounds = np.searchsorted(fidx, np.arange(len(group) + 1))          # fidx is sorted (recording-major output)
    blocks = []
    for f, (clip_id, _) in enumerate(group):
        a, b = int(bounds[f]), int(bounds[f + 1])
        k = keep[a:b]
        blocks.append((clip_id, onset[a:b][k], offset[a:b][k], duration[a:b][k], lab[a:b][k]))
    return blocks


def run_group(engine, group, lo, hi, labels, batch_size):
    """One batched forward + one segmented decode, synchronously -> one block per clip."""
    logits, file_rows = launch_group(engine, group, batch_size)
    engine.decode_launch(logits, file_rows, lo, hi)
    return collect_group(engine, group)


def main(dataset: str, config: str = "VTC-2.0/model/config.yml", checkpoint: str = "VTC-2.0/model/best.ckpt",
         threshold: float = 0.5, min_duration_on_s: float = 0.1, min_duration_off_s: float = 0.3, batch_size: int = 0,
         device: str = "cuda", array_id: int | None = None, array_count: int | None = None,
         per_clip: bool = False) -> None:
    set_seeds(42)
    if device != "cuda":
        raise SystemExit(f"vtc_amd runs on ROCm GPUs only (--device {device} requested); it has no CPU path")
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm GPU visible; vtc_amd has no CPU path")
    paths = get_dataset_paths(dataset)
    shard_dir = paths.output / "shards"
    out_dir = paths.output / "vtc_clips"
    shard_id = array_id if array_id is not None else 0
    logger.info(f"Dataset : {dataset}")
    logger.info(f"Shards  : {shard_dir}")
    logger.info(f"Output  : {out_dir}")

    tar_files = sorted(shard_dir.glob("*.tar"))
    if not tar_files:
        logger.error(f"No .tar files found in {shard_dir}")
        sys.exit(1)
    logger.info(f"Found {len(tar_files)} tar shards")
    if array_id is not None and array_count is not None:
        tar_files = shard_list(tar_files, array_id, array_count)
        logger.info(f"Shard {array_id}/{array_count - 1}: {len(tar_files)} tars")

    # ---- resume: clip_ids already present in ANY shard's raw parquet
    raw_dir, merged_dir = out_dir / "vtc_raw", out_dir / "vtc_merged"
    raw_path = raw_dir / f"shard_{shard_id}.parquet"
    completed: set = set()
    if raw_dir.exists():
        for f in sorted(raw_dir.glob("shard_*.parquet")):
            completed.update(pq.read_table(f, columns=["clip_id"]).column("clip_id").to_pylist())
        if completed:
            logger.info(f"Resume: {len(completed)} clips already done")

    # ---- model
    if Path(config).exists():
        model_config: Config = load_config(config)
    else:
        logger.warning(f"config {config} not found; using built-in defaults (4 s windows, VTC_LABELS order)")
        model_config = Config()
    l_encoder = MultiLabelEncoder(labels=model_config.data.classes)
    model = Models[model_config.model.name].load_from_checkpoint(
        checkpoint_path=checkpoint, label_encoder=l_encoder, config=model_config, train=False)
    model.eval()
    dev_index = torch.cuda.current_device()
    model.to(torch.device("cuda", dev_index))
    chunk_duration_s = model_config.audio.chunk_duration_s
    if batch_size <= 0:
        batch_size = tile_friendly_batch(max(16, int(round(164 * 4.0 / float(chunk_duration_s)))), float(chunk_duration_s))
    engine = model.engine(chunk_duration_s, batch_size)
    labels = l_encoder.labels
    lo, hi = logit_bounds({l: {"lower_bound": threshold, "upper_bound": 1.0} for l in labels}, labels)
    step = engine.step

    seg_blocks: list = []
    n_processed = n_skipped = n_errors = 0
    t0 = last_log_t = time.time()
    group, group_windows = [], 0
    pending = None            # (group, logits): forward and decode enqueued, not collected yet

    def redo(items, e):
        nonlocal n_processed, n_errors
        logger.warning(f"batch of {len(items)} clips failed ({e}); retrying one by one")
        engine.decode_abandon()
        for item in items:       # a failing batch is retried clip by clip so one bad clip cannot take the others down
            try:
                seg_blocks.extend(run_group(engine, [item], lo, hi, labels, batch_size))
                n_processed += 1
            except Exception as e1:
                logger.warning(f"{item[0]}: {e1}")
                n_errors += 1

    def drain():
        nonlocal pending, n_processed
        if pending is None:
            return
        (items, _logits), pending = pending, None
        try:
            seg_blocks.extend(collect_group(engine, items))
            n_processed += len(items)
        except Exception as e:
            redo(items, e)

    def flush():
        """forward(i+1), collect(i), decode_launch(i+1): the GPU already runs the next batch while this thread turns
        the previous one into rows (per_clip restores the reference's one clip at a time, fully synchronous)."""
        nonlocal group, group_windows, pending
        if not group:
            return
        items, group, group_windows = group, [], 0
        try:
            logits, file_rows = launch_group(engine, items, batch_size)
        except Exception as e:
            drain()
            redo(items, e)
            return
        drain()
        try:
            engine.decode_launch(logits, file_rows, lo, hi)
            pending = (items, logits)
        except Exception as e:
            redo(items, e)
        if per_clip:
            drain()

    # Window packing (pipeline/vtc.py::pack_forward): clips share forwards window by window, so every forward but the last
    # is full.  Grouping whole clips until a batch is reached overshoots it by up to a clip, and the overshoot (a few
    # windows) then ran as a forward of its own.
    carry, carry_windows = deque(), 0

    def pump(final: bool):
        nonlocal carry_windows, pending
        while carry_windows >= batch_size or (final and carry_windows):
            nw = min(batch_size, carry_windows)
            touched = [(o.i, o.wav) for o in carry]
            try:
                done, out = pack_forward(engine, carry, nw)
            except Exception as e:
                carry.clear()
                carry_windows = 0
                drain()
                redo(touched, e)
                return
            carry_windows -= nw
            drain()
            if done:
                items = [(o.i, o.wav) for o in done]
                try:
                    logits, file_rows, _ = packed_decode_input(done, out)
                    engine.decode_launch(logits, file_rows, lo, hi)
                    pending = (items, logits)
                except Exception as e:
                    redo(items, e)

    # Reader thread: tar members -> PCM in memory (FLAC through the library's decoder, outside the GIL) -> waveforms in HBM, a
    # bounded queue ahead of the GPU loop.  Order is the tar order, as in the reference.  The staging is bundled as in the file
    # loop (pipeline/vtc.py::stage_bundle): consecutive mono 16 kHz int16 clips go back to back into one pinned buffer, cross
    # PCIe in ONE copy and are converted by ONE front-end launch on this thread's stream (int16 -> float32 is per sample, so
    # the clips are bit-identical to clips staged one by one); their waveforms are views of one device array, which the
    # packed forward uses as it is.  One pageable copy + one launch + one synchronisation per clip on the main thread was what
    # bounded the clip loop (0.5 ms per clip).  Any other clip (stereo, another rate, float) is staged on its own, here too.
    todo: queue.Queue = queue.Queue(maxsize=4 * max(batch_size, 16))
    stop = threading.Event()
    bundle_cap = 8 << 20                      # int16 samples per bundle (16 MB of pinned memory)

    def reader():
        torch.cuda.set_device(engine.device)
        copy_stream = torch.cuda.Stream(engine.device)
        host = torch.empty(bundle_cap, dtype=torch.int16, pin_memory=True)
        host_np = host.numpy()
        bundle, pos, windows = [], 0, 0       # (clip_id, first sample, samples)

        def emit():
            nonlocal bundle, pos, windows
            if not bundle:
                return
            items, n_total, bundle, pos, windows = bundle, pos, [], 0, 0
            try:
                with torch.cuda.stream(copy_stream):
                    arena = engine.prepare_audio(host[:n_total].to(engine.device, non_blocking=True), 16000)
                copy_stream.synchronize()     # the pinned buffer is free again and the waveforms are ready
            except Exception as e:
                for clip_id, _, _ in items:
                    todo.put((clip_id, None, e))
                return
            for clip_id, a, n in items:
                todo.put((clip_id, arena[a:a + n], None))

        def stage_one(pcm, sr):
            with torch.cuda.stream(copy_stream):
                wav = engine.prepare_audio(torch.from_numpy(pcm).to(engine.device), sr)
            copy_stream.synchronize()
            return wav

        def handle(clip_id, fut):
            """One clip, in tar order: a resumed clip (fut None), or its decoded PCM (or the decoder's exception)."""
            nonlocal pos, windows
            if fut is None:
                emit()
                todo.put((clip_id, None, None))
                return
            try:
                pcm, sr = fut.result()
                plain = (not per_clip and pcm.dtype == np.int16 and pcm.ndim == 1 and sr == 16000
                         and RF_SIZE <= pcm.shape[0] <= bundle_cap)
                if plain:
                    n = int(pcm.shape[0])
                    n_win = -(-(n - RF_SIZE + 1) // step)
                    if pos + n > bundle_cap or (bundle and windows + n_win > batch_size):
                        emit()
                    host_np[pos:pos + n] = pcm
                    bundle.append((clip_id, pos, n))
                    pos += n
                    windows += n_win
                    if windows >= batch_size:
                        emit()
                else:
                    emit()
                    todo.put((clip_id, stage_one(pcm, sr), None))
            except Exception as e:
                emit()
                todo.put((clip_id, None, e))

        # The members are decoded on a small pool (the packer writes FLAC by default, src/packaging/writer.py:36; the library's
        # decoder runs outside the GIL at ~2 audio-hours/s per thread, an eighth of what the GPU takes): this thread walks the
        # tar, submits the decodes and consumes them in tar order, a bounded number of clips ahead.
        n_dec = max(1, int(os.environ.get("VTC_CLIP_DECODERS", str(max(1, min(8, len(os.sched_getaffinity(0)) - 2))))))
        ahead: deque = deque()
        try:
            with ThreadPoolExecutor(max_workers=n_dec, thread_name_prefix="clip-decode") as pool:
                for tar_path in tar_files:
                    for sample in iter_tar_samples(tar_path):
                        if stop.is_set():
                            return
                        clip_id = sample["__key__"]
                        ahead.append((clip_id, None if clip_id in completed else pool.submit(decode_clip, sample)))
                        while len(ahead) > 4 * n_dec:
                            handle(*ahead.popleft())
                while ahead and not stop.is_set():
                    handle(*ahead.popleft())
            emit()
        except BaseException as e:          # a broken tar must not leave the consumer waiting
            todo.put((None, None, e))
        finally:
            todo.put(None)

    reader_thread = threading.Thread(target=reader, name="clip-reader", daemon=True)
    reader_thread.start()
    try:
        while True:
            item = todo.get()
            if item is None:
                break
            clip_id, wav, err = item
            if clip_id is None:
                raise err
            if err is None and wav is None:
                n_skipped += 1
                continue
            try:
                if err is not None:
                    raise err
                if wav.numel() < RF_SIZE:
                    raise ValueError(f"audio too short: {wav.numel()} samples < receptive field {RF_SIZE}")
            except Exception as e:
                logger.warning(f"{clip_id}: {e}")
                n_errors += 1
                continue
            n_win = max(1, -(-(int(wav.numel()) - RF_SIZE + 1) // step))
            if per_clip:
                group.append((clip_id, wav))
                flush()
            else:
                carry.append(OpenRecording(clip_id, wav, n_win))
                carry_windows += n_win
                pump(False)
            now = time.time()
            if now - last_log_t >= 60:
                rate = n_processed / max(now - t0, 1e-9)
                logger.info(f"  {n_processed} processed, {n_skipped} skipped, {n_errors} errors  ({rate:.1f} clips/s)")
                last_log_t = now
        flush()
        pump(True)
        drain()
    finally:
        stop.set()
        while reader_thread.is_alive():     # unblock a reader waiting on the full queue
            try:
                todo.get(timeout=0.05)
            except queue.Empty:
                pass

    new_seg = clips_table(seg_blocks, labels)
    if raw_path.exists() and completed:
        prev = pq.read_table(raw_path).cast(CLIP_SCHEMA)
        new_ids = pa.array(sorted({b[0] for b in seg_blocks if len(b[1])}), pa.string())
        kept = prev.filter(pc.invert(pc.is_in(prev.column("clip_id"), value_set=new_ids))) if len(new_ids) else prev
        parts = [t for t in (kept, new_seg) if t.num_rows]
        seg_df = pa.concat_tables(parts) if parts else new_seg
    else:
        seg_df = new_seg
    atomic_write_parquet(seg_df, raw_path)
    merged = merge_segments(seg_df.rename_columns(["uid", "onset", "offset", "duration", "label"]), min_duration_off_s,
                            min_duration_on_s).rename_columns(["clip_id", "onset", "offset", "duration", "label"])
    atomic_write_parquet(merged, merged_dir / f"shard_{shard_id}.parquet")

    wall = time.time() - t0
    logger.info("-" * 50)
    logger.info(f"Shard {shard_id} complete")
    logger.info(f"  Processed : {n_processed} clips  ({n_errors} errors, {n_skipped} skipped)")
    logger.info(f"  Segments  : {seg_df.num_rows:,} raw, {merged.num_rows:,} merged")
    logger.info(f"  Wall time : {hhmmss(wall)}   ({n_processed / max(wall, 1e-9):.1f} clips/s)")
    logger.info("-" * 50)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="VTC inference on clips from WebDataset shards (MI355X-native), "
                                            "storing clip-relative segments for alignment analysis.")
    p.add_argument("dataset", help="Dataset name (e.g. seedlings_10)")
    p.add_argument("--config", default="VTC-2.0/model/config.yml", help="model config YAML")
    p.add_argument("--checkpoint", default="VTC-2.0/model/best.ckpt", help="model checkpoint or 'synthetic[:seed]'")
    p.add_argument("--threshold", type=float, default=0.5, help="Sigmoid threshold (default: 0.5)")
    p.add_argument("--min_duration_on_s", type=float, default=0.1, help="Remove segments shorter than this (0.1 s)")
    p.add_argument("--min_duration_off_s", type=float, default=0.3, help="Merge same-label gaps smaller than this (0.3 s)")
    p.add_argument("--batch_size", type=int, default=0, help="Windows per forward pass, shared by the clips of a batch; 0 = 164 windows of 4 s, scaled by the window length and trimmed to whole rounds of GEMM tiles (packing.py::tile_friendly_batch)")
    p.add_argument("--device", default="cuda", choices=["cuda", "cpu", "mps"], help="Only 'cuda' (ROCm) is supported")
    p.add_argument("--array_id", type=int, help="Shard index (SLURM array task ID or launcher rank)")
    p.add_argument("--array_count", type=int, help="Number of shards")
    p.add_argument("--per_clip", action="store_true", help="one forward + decode per clip, as the reference does (for A/B timing)")
    return p


if __name__ == "__main__":
    main(**vars(build_parser().parse_args()))
