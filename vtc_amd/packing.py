"""Window packing: recordings share forward passes window by window (host logic above vtc_forward_windows), and the batch
size at which those forwards fill whole rounds of GEMM tiles.

Used by the file loop (pipeline/vtc.py) and by the clips caller (analysis/vtc_on_clips.py).  The reference runs one recording
per `apply_model_on_audio` call (src/pipeline/vtc.py:236-243); its window plan per recording is kept exactly
(src/analysis/chunk_artifact.py:60-83), only which forward a window rides in changes.
"""

from __future__ import annotations

from collections import deque

import numpy as np
import torch

from .engine import RF_SIZE, RF_STEP


class OpenRecording:
    """A staged recording some of whose windows have not run yet (`pack_forward`)."""
    __slots__ = ("i", "wav", "n", "n_win", "next_w", "parts")

    def __init__(self, i: int, wav: torch.Tensor, n_win: int):
        self.i, self.wav, self.n, self.n_win, self.next_w, self.parts = i, wav, int(wav.numel()), n_win, 0, []


def pack_forward(engine, carry: deque, n_windows: int):
    """One forward over the next n_windows windows of the recordings queued in `carry`, whatever recording they belong to:
    the last windows of one recording, whole recordings, and the first windows of another may share it, so that every
    forward but a run's last is FULL.  (A 600 s recording is 61 windows of 10 s: alone it leaves 3 of a 64-window batch
    empty, and twenty-two of them fit 21 forwards.)  Windows are independent (a frame's logits do not depend on its
    batch mates: tests/test_gpu_parity.py::test_full_size_batch64_10s_properties), so the rows are bit-identical to those
    of the recording run alone.  The window plan per recording is vtc_infer_waveform's (step T * 320, zero-padded tail,
    frames whose receptive field lies in the audio).  Enqueues only; returns (recordings completed by this forward, each with
    its stitched logits in `parts`, in queue order; the forward's logits buffer)."""
    W, T, step, nl = engine.window_samples, engine.frames_per_window, engine.step, engine.n_labels
    cur = torch.cuda.current_stream(engine.device) if torch.device(engine.device).type == "cuda" else None
    # plain Python lists, converted once: with thousands of 1-10 s recordings per second this runs per recording, under the GIL
    # the reader threads and the row-building thread also need
    segs, runs, start, valid, keep, dst = [], [], [], [], [], []
    need, base, rows = n_windows, 0, 0
    while need and carry:
        o = carry[0]
        w0, n = o.next_w, o.n
        cnt = min(need, o.n_win - w0)
        o.next_w = w0 + cnt
        need -= cnt
        if o.next_w == o.n_win:
            carry.popleft()
        a, b = w0 * step, min(n, (w0 + cnt - 1) * step + W)
        if w0 == 0 and cur is not None:
            o.wav.record_stream(cur)          # staged on a reader thread's stream
        piece = o.wav if (a == 0 and b == n) else o.wav[a:b]
        # pieces that follow each other in one device array (a bundle of short files, stage_bundle) are one run: no copy
        # (adjacent blocks of different allocations do not count: the storage must be the same)
        ptr, store = piece.data_ptr(), piece.untyped_storage().data_ptr()
        if runs and runs[-1][2] == ptr and runs[-1][3] == store:
            runs[-1][1] += b - a
            runs[-1][2] += 4 * (b - a)
        else:
            runs.append([piece, b - a, ptr + 4 * (b - a), store])
        off, r0 = base - a, rows
        for w in range(w0, w0 + cnt):
            s0 = w * step
            rem = n - s0
            k = (rem - RF_SIZE) // RF_STEP + 1
            k = k if k < T else T
            start.append(off + s0)
            valid.append(rem if rem < W else W)
            keep.append(k)
            dst.append(rows)
            rows += k
        segs.append((o, r0, rows))
        base += b - a
    pieces = [r[0] if r[1] == r[0].numel() else torch.as_strided(r[0], (r[1],), (1,), r[0].storage_offset()) for r in runs]
    # several runs: one copy of <= a batch of windows (41 MB at 64 x 10 s, ~20 us) in front of a 10 ms forward
    arena = pieces[0] if len(pieces) == 1 else torch.cat(pieces)
    out = torch.empty((rows, nl), dtype=torch.float32, device=engine.device)
    with torch.no_grad():
        engine.forward_windows(arena, np.array(start, dtype=np.int64), np.array(valid, dtype=np.int32),
                               np.array(dst, dtype=np.int64), np.array(keep, dtype=np.int32), out)
    done = []
    for o, r0, r1 in segs:
        o.parts.append((out, r0, r1))         # sliced only when needed (packed_decode_input)
        if o.next_w == o.n_win:
            done.append(o)
    return done, out


def packed_decode_input(done: list, out: torch.Tensor):
    """(logits, file_rows, n_samples) of the recordings a forward completed, in the form gpu_decode_launch takes: the rows of
    the forward's own buffer when every recording lies inside it (they are consecutive: only a forward's first recording
    can have earlier parts and only its last one can be incomplete), otherwise one concatenation."""
    rows = np.zeros(len(done) + 1, dtype=np.int64)
    single = True
    for k, o in enumerate(done):
        rows[k + 1] = rows[k] + sum(r1 - r0 for _, r0, r1 in o.parts)
        single = single and len(o.parts) == 1
    if single:
        a = done[0].parts[0][1]
        logits = out[a:a + int(rows[-1])]
    else:
        logits = torch.cat([t[r0:r1] for o in done for t, r0, r1 in o.parts])
    return logits, rows, [o.n for o in done]


def forward_cost_per_window(batch: int, chunk_duration_s: float, n_cus: int = 256):
    """Relative GEMM time of one forward divided by its windows: rounds of 256-row tiles over n_cus CUs x (K x tile width), summed
    over the GEMMs of csrc/vtc_api.hip's dispatch (conv1-6 and FFN1 on 256-wide tiles, projection / QKV / out-projection / FFN2
    on 192-wide ones).  None when the window is shorter than the receptive field."""
    window = int(round(chunk_duration_s * 16000))
    lens, n = [], window
    for k, s in ((10, 5), (3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2)):
        n = (n - k) // s + 1
        lens.append(n)
    if batch < 1 or lens[-1] < 1:
        return None
    frames = lens[-1]

    def rounds(tiles: int) -> int:
        return -(-tiles // n_cus)

    t = 0.0
    for i, k in enumerate((1536, 1536, 1536, 1536, 1024, 1024)):
        t += rounds(-(-batch * lens[i + 1] // 256) * 2) * k * 256
    tm = -(-batch * frames // 256)
    t += rounds(tm * 4) * 512 * 192
    t += 12 * (rounds(tm * 12) * 768 * 192 + rounds(tm * 12) * 768 * 256 + rounds(tm * 4) * 768 * 192 + rounds(tm * 4) * 3072 * 192)
    return t / batch


def tile_friendly_batch(max_batch: int, chunk_duration_s: float, n_cus: int = 256) -> int:
    """The batch size <= max_batch (and >= 80 % of it) at which the persistent GEMMs waste the least of their last round of
    tiles.  Every GEMM of the forward walks 256-row tiles over the card's 256 CUs, so a launch costs ceil(tiles / 256) rounds
    whatever the last round holds: at 4 s windows a batch of 128 is 100 row tiles, the out-projection and FFN2 (4 column
    tiles each) then run 400 tiles = 1.56 rounds in the time of 2, and 164 windows (128 row tiles, 512 tiles = 2 rounds
    exactly) go through in barely more time.  Measured hot-path rates (tools/batch_sweep.py, profiles/r03_batch_friendly.json):
    4 s windows 16.2 audio-h/s at 128, 17.3 at 164, 16.5 at 512, 17.3 at 493; 10 s windows 16.9 at 64, 16.1 at 208, 16.9 at 197.
    The candidates are ranked by forward_cost_per_window (rounds x tile work per window).
    Since window packing every forward but a run's last holds exactly this many windows."""
    if max_batch < 8 or forward_cost_per_window(max_batch, chunk_duration_s, n_cus) is None:
        return max_batch
    ref = forward_cost_per_window(max_batch, chunk_duration_s, n_cus)
    lo = max(1, int(max_batch * 0.8))
    return min(range(max_batch, lo - 1, -1), key=lambda b: (round(forward_cost_per_window(b, chunk_duration_s, n_cus) / ref, 4), -b))
