"""VtcEngine -- the Python owner of one `vtc_model` handle (one per GPU / process).

PyTorch is plumbing here: device buffers, streams and the tensors handed back to the host code.
All arithmetic of the hot path runs in libvtc_amd.so (hand-written gfx950 kernels).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as L

SAMPLE_RATE = 16_000
RF_STEP = 320   # reference: src/analysis/chunk_artifact.py:51
RF_SIZE = 400   # reference: src/analysis/chunk_artifact.py:11-13
DTYPES = {"bf16": 0, "f16": 1}

import weakref

_LIVE: "weakref.WeakSet[VtcEngine]" = weakref.WeakSet()   # engines with an open handle (segma_compat resolves foreign tensors)
_SERIAL = [0]


def live_engines(n_labels: int | None = None, device=None) -> list:
    """Open engines, newest first, optionally only those with this label count / on this device."""
    out = [e for e in _LIVE if getattr(e, "_handle", None) and e._handle.value]
    if n_labels is not None:
        out = [e for e in out if e.n_labels == int(n_labels)]
    if device is not None and torch.device(device).type == "cuda":
        d = torch.device(device)
        out = [e for e in out if d.index is None or e.device.index == d.index]
    return sorted(out, key=lambda e: -e._serial)


def frames_for_samples(n_samples: int) -> int:
    return 0 if n_samples < RF_SIZE else (n_samples - RF_SIZE) // RF_STEP + 1


def _np32(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        t = t.detach().to("cpu", torch.float32).numpy()
    return np.ascontiguousarray(np.asarray(t, dtype=np.float32))


def fold_weight_norm(sd: dict) -> np.ndarray:
    """Positional conv weight with weight_norm(dim=2) folded: W = g * v / ||v|| over dims (0,1)
    (HF modeling_hubert.py:60-78).  Accepts the parametrised, the legacy (_g/_v) or a plain weight."""
    p = "encoder.pos_conv_embed.conv."
    if p + "parametrizations.weight.original1" in sd:
        g, v = _np32(sd[p + "parametrizations.weight.original0"]), _np32(sd[p + "parametrizations.weight.original1"])
    elif p + "weight_v" in sd:
        g, v = _np32(sd[p + "weight_g"]), _np32(sd[p + "weight_v"])
    else:
        return _np32(sd[p + "weight"])
    norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(0, 1), keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(g * v / norm)


class VtcEngine:
    def __init__(self, state_dict: dict, n_labels: int, chunk_duration_s: float, max_batch: int = 64,
                 mfma_dtype: str = "f16", device: int | str | torch.device = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("vtc_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
        if mfma_dtype not in DTYPES:
            raise ValueError(f"mfma_dtype must be one of {list(DTYPES)}")
        self._lib = L.load()
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise RuntimeError(f"vtc_amd runs on ROCm GPUs only, got device {self.device}")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device(f"cuda:{dev_index}")
        self.n_labels = int(n_labels)
        self.window_samples = int(round(chunk_duration_s * SAMPLE_RATE))
        self.max_batch = int(max_batch)
        self.mfma_dtype = mfma_dtype
        self._handle = C.c_void_p()

        keep = []  # numpy arrays must outlive vtc_create

        def ptr(name_or_arr):
            a = _np32(state_dict[name_or_arr]) if isinstance(name_or_arr, str) else _np32(name_or_arr)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_float))

        w = L.VtcWeights()
        for i in range(7):
            w.conv_w[i] = ptr(f"feature_extractor.conv_layers.{i}.conv.weight")
        w.gn_w, w.gn_b = ptr("feature_extractor.conv_layers.0.layer_norm.weight"), ptr("feature_extractor.conv_layers.0.layer_norm.bias")
        w.fp_ln_w, w.fp_ln_b = ptr("feature_projection.layer_norm.weight"), ptr("feature_projection.layer_norm.bias")
        w.fp_w, w.fp_b = ptr("feature_projection.projection.weight"), ptr("feature_projection.projection.bias")
        w.pos_w, w.pos_b = ptr(fold_weight_norm(state_dict)), ptr("encoder.pos_conv_embed.conv.bias")
        w.enc_ln_w, w.enc_ln_b = ptr("encoder.layer_norm.weight"), ptr("encoder.layer_norm.bias")
        for l in range(L.N_LAYERS):
            p = f"encoder.layers.{l}."
            lw = w.layer[l]
            for short, long in (("q", "attention.q_proj"), ("k", "attention.k_proj"), ("v", "attention.v_proj"),
                                ("o", "attention.out_proj"), ("ff1", "feed_forward.intermediate_dense"),
                                ("ff2", "feed_forward.output_dense"), ("ln1", "layer_norm"), ("ln2", "final_layer_norm")):
                setattr(lw, short + "_w", ptr(p + long + ".weight"))
                setattr(lw, short + "_b", ptr(p + long + ".bias"))
        hw = _np32(state_dict["head.weight"])
        if hw.shape != (self.n_labels, 768):
            raise ValueError(f"head.weight has shape {hw.shape}, expected ({self.n_labels}, 768)")
        w.head_w, w.head_b = ptr(hw), ptr("head.bias")

        spec = L.VtcSpec(abi_version=L.ABI_VERSION, n_labels=self.n_labels, window_samples=self.window_samples,
                         max_batch=self.max_batch, mfma_dtype=DTYPES[mfma_dtype])
        L.check(self._lib.vtc_create(C.byref(spec), C.byref(w), dev_index, C.byref(self._handle)), None)
        self.frames_per_window = int(self._lib.vtc_frames_per_window(self._handle))
        self.step = self.frames_per_window * RF_STEP   # reference: chunk_artifact.py:71-73
        _SERIAL[0] += 1
        self._serial = _SERIAL[0]
        _LIVE.add(self)

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_handle", None) and self._handle.value:
            self._lib.vtc_destroy(self._handle)
            self._handle = C.c_void_p()
            _LIVE.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_wav(self, wav: torch.Tensor):
        if wav.device != self.device or wav.dtype != torch.float32 or not wav.is_contiguous() or wav.dim() != 1:
            raise ValueError("waveform must be a contiguous 1-D float32 tensor on the engine's device")

    # ------------------------------------------------------------------ forward
    def forward_windows(self, wav: torch.Tensor, start, valid, dst_row, keep, logits: torch.Tensor) -> None:
        """Run one batch of windows (include/vtc_amd.h: vtc_forward_windows)."""
        self._check_wav(wav)
        start = np.ascontiguousarray(start, dtype=np.int64)
        valid = np.ascontiguousarray(valid, dtype=np.int32)
        dst_row = np.ascontiguousarray(dst_row, dtype=np.int64)
        keep = np.ascontiguousarray(keep, dtype=np.int32)
        n = len(start)
        if not (len(valid) == len(dst_row) == len(keep) == n):
            raise ValueError("window arrays differ in length")
        if n and int((start + valid).max()) > wav.numel():
            raise ValueError("window reaches past the end of the waveform arena")
        if logits.device != self.device or logits.dtype != torch.float32 or not logits.is_contiguous():
            raise ValueError("logits must be a contiguous float32 tensor on the engine's device")
        if n and int((dst_row + keep).max()) * self.n_labels > logits.numel():
            raise ValueError("logits buffer too small for dst_row + keep")
        L.check(self._lib.vtc_forward_windows(
            self._handle, C.c_void_p(wav.data_ptr()), start.ctypes.data_as(L._i64p), valid.ctypes.data_as(L._i32p),
            dst_row.ctypes.data_as(L._i64p), keep.ctypes.data_as(L._i32p), n, C.c_void_p(logits.data_ptr()),
            self._stream()), self._handle)

    def infer_waveform(self, wav: torch.Tensor, batch_size: int | None = None) -> torch.Tensor:
        """Stitched logits (n_frames, n_labels) of one waveform (include/vtc_amd.h: vtc_infer_waveform)."""
        self._check_wav(wav)
        n = int(wav.numel())
        rows = frames_for_samples(n)
        logits = torch.empty((max(rows, 1), self.n_labels), dtype=torch.float32, device=self.device)
        nf = C.c_int64(0)
        L.check(self._lib.vtc_infer_waveform(self._handle, C.c_void_p(wav.data_ptr()), n,
                                             int(batch_size or self.max_batch), C.c_void_p(logits.data_ptr()),
                                             C.byref(nf), self._stream()), self._handle)
        return logits[: nf.value]

    # ------------------------------------------------------------------ several recordings per call
    def infer_waveforms(self, wavs, batch_size: int | None = None):
        """List of 1-D float32 device waveforms (e.g. packaged clips) -> (logits, file_rows): stitched logits of
        recording f are logits[file_rows[f]:file_rows[f+1]] (include/vtc_amd.h: vtc_infer_waveforms)."""
        if not wavs:
            raise ValueError("no waveforms")
        for w in wavs:
            self._check_wav(w)
        n = np.asarray([int(w.numel()) for w in wavs], dtype=np.int64)
        if len(wavs) == 1:
            arena = wavs[0]
        else:
            # recordings staged as one bundle are consecutive views of one device array: use it as the arena, no copy
            ptr = np.asarray([w.data_ptr() for w in wavs], dtype=np.int64)
            store = wavs[0].untyped_storage().data_ptr()
            if bool(np.all(ptr[1:] == ptr[:-1] + 4 * n[:-1])) and all(w.untyped_storage().data_ptr() == store for w in wavs):
                arena = torch.as_strided(wavs[0], (int(n.sum()),), (1,), wavs[0].storage_offset())
            else:
                arena = torch.cat(wavs)
        off = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int64)
        rows = int(sum(frames_for_samples(int(k)) for k in n))
        logits = torch.empty((max(rows, 1), self.n_labels), dtype=torch.float32, device=self.device)
        file_rows = np.zeros(len(wavs) + 1, dtype=np.int64)
        L.check(self._lib.vtc_infer_waveforms(self._handle, C.c_void_p(arena.data_ptr()), off.ctypes.data_as(L._i64p),
                                              n.ctypes.data_as(L._i64p), len(wavs), int(batch_size or self.max_batch),
                                              C.c_void_p(logits.data_ptr()), file_rows.ctypes.data_as(L._i64p),
                                              self._stream()), self._handle)
        arena.record_stream(torch.cuda.current_stream(self.device))
        return logits[: int(file_rows[-1])], file_rows

    def decode_files(self, logits: torch.Tensor, file_rows, lo_logit, hi_logit, summary: bool = False):
        """Threshold + run-length decode of several recordings' rows; runs never cross a recording boundary.
        Returns (file_index[int32], start_samples[int64], end_samples[int64], label_index[int32]), samples relative
        to the recording (include/vtc_amd.h: vtc_decode_files); with summary=True also (max_logit[float32],
        mean_sigmoid[float64]) per recording."""
        if logits.device != self.device or logits.dtype != torch.float32 or not logits.is_contiguous():
            raise ValueError("logits must be a contiguous float32 tensor on the engine's device")
        file_rows = np.ascontiguousarray(file_rows, dtype=np.int64)
        n = int(file_rows[-1])
        if n > int(logits.shape[0]):
            raise ValueError("file_rows reach past the logits buffer")
        lo = np.ascontiguousarray(lo_logit, dtype=np.float32)
        hi = np.ascontiguousarray(hi_logit, dtype=np.float32)
        if len(lo) != self.n_labels or len(hi) != self.n_labels:
            raise ValueError("need one lower and one upper bound per label")
        cap = max(1024, (n // 2 + len(file_rows)) * self.n_labels)
        fidx = np.empty(cap, dtype=np.int32)
        s = np.empty(cap, dtype=np.int64)
        e = np.empty(cap, dtype=np.int64)
        lab = np.empty(cap, dtype=np.int32)
        n_out = C.c_int64(0)
        nf = len(file_rows) - 1
        mx = np.empty(nf, dtype=np.float32) if summary else None
        mean = np.empty(nf, dtype=np.float64) if summary else None
        L.check(self._lib.vtc_decode_files(self._handle, C.c_void_p(logits.data_ptr()), file_rows.ctypes.data_as(L._i64p),
                                           nf, lo.ctypes.data_as(L._f32p), hi.ctypes.data_as(L._f32p),
                                           fidx.ctypes.data_as(L._i32p), s.ctypes.data_as(L._i64p), e.ctypes.data_as(L._i64p),
                                           lab.ctypes.data_as(L._i32p), cap, C.byref(n_out),
                                           mx.ctypes.data_as(L._f32p) if summary else None,
                                           mean.ctypes.data_as(L._f64p) if summary else None, self._stream()), self._handle)
        k = n_out.value
        out = (fidx[:k].copy(), s[:k].copy(), e[:k].copy(), lab[:k].copy())
        return out + (mx, mean) if summary else out

    def decode_launch(self, logits: torch.Tensor, file_rows, lo_logit, hi_logit, summary: bool = False) -> None:
        """First half of decode_files (include/vtc_amd.h: vtc_decode_launch): enqueue the decode of these rows on the
        current stream and return at once; more work (the next forward) may be enqueued before decode_collect().
        file_rows = [0, n_frames] decodes one recording.  `logits` must stay alive until decode_collect()."""
        if logits.device != self.device or logits.dtype != torch.float32 or not logits.is_contiguous():
            raise ValueError("logits must be a contiguous float32 tensor on the engine's device")
        file_rows = np.ascontiguousarray(file_rows, dtype=np.int64)
        if int(file_rows[-1]) > int(logits.shape[0]):
            raise ValueError("file_rows reach past the logits buffer")
        lo = np.ascontiguousarray(lo_logit, dtype=np.float32)
        hi = np.ascontiguousarray(hi_logit, dtype=np.float32)
        if len(lo) != self.n_labels or len(hi) != self.n_labels:
            raise ValueError("need one lower and one upper bound per label")
        L.check(self._lib.vtc_decode_launch(self._handle, C.c_void_p(logits.data_ptr()), file_rows.ctypes.data_as(L._i64p),
                                            len(file_rows) - 1, lo.ctypes.data_as(L._f32p), hi.ctypes.data_as(L._f32p),
                                            int(summary), self._stream()), self._handle)
        self._pending_decode = (len(file_rows) - 1, bool(summary), logits)

    def decode_collect(self):
        """Second half: wait for the launched decode (not for anything enqueued after it) and return what decode_files
        returns: (file_index, start_samples, end_samples, label_index[, max_logit, mean_sigmoid])."""
        if getattr(self, "_pending_decode", None) is None:
            raise RuntimeError("decode_collect() without decode_launch()")
        nf, summary, _keep = self._pending_decode
        mx = np.empty(nf, dtype=np.float32) if summary else None
        mean = np.empty(nf, dtype=np.float64) if summary else None
        n_out = C.c_int64(0)
        args_tail = (mx.ctypes.data_as(L._f32p) if summary else None, mean.ctypes.data_as(L._f64p) if summary else None)
        rc = self._lib.vtc_decode_collect(self._handle, None, None, None, None, 0, C.byref(n_out), *args_tail)
        k = n_out.value
        fidx, s = np.empty(k, dtype=np.int32), np.empty(k, dtype=np.int64)
        e, lab = np.empty(k, dtype=np.int64), np.empty(k, dtype=np.int32)
        if rc == L.ERR_CAPACITY:          # the size is known now: second call copies the runs out
            rc = self._lib.vtc_decode_collect(self._handle, fidx.ctypes.data_as(L._i32p), s.ctypes.data_as(L._i64p),
                                              e.ctypes.data_as(L._i64p), lab.ctypes.data_as(L._i32p), k, C.byref(n_out), *args_tail)
        self._pending_decode = None
        L.check(rc, self._handle)
        out = (fidx, s, e, lab)
        return out + (mx, mean) if summary else out

    def decode_abandon(self) -> None:
        """Drop a launched decode whose results are no longer wanted (error paths), so that the next decode_launch()
        is accepted."""
        if getattr(self, "_pending_decode", None) is not None:
            try:
                self.decode_collect()
            except Exception:
                self._pending_decode = None

    # ------------------------------------------------------------------ audio front-end
    def prepare_audio(self, pcm: torch.Tensor, sample_rate: int) -> torch.Tensor:
        """Interleaved PCM on the device, int16 or float32, shape (n,) or (n, channels) -> mono float32 at 16 kHz
        (include/vtc_amd.h: vtc_prepare_audio).  Packed 24-bit PCM is a uint8 tensor of shape (n, channels, 3), the bytes
        as a WAV file stores them.  Runs on the current stream."""
        if pcm.dtype == torch.uint8:
            if pcm.device != self.device or not pcm.is_contiguous() or pcm.dim() != 3 or pcm.shape[2] != 3:
                raise ValueError("packed 24-bit pcm must be a contiguous uint8 (n, channels, 3) tensor on the engine's device")
        elif pcm.device != self.device or not pcm.is_contiguous() or pcm.dim() not in (1, 2):
            raise ValueError("pcm must be a contiguous (n,) or (n, channels) tensor on the engine's device")
        if pcm.dtype not in (torch.int16, torch.float32, torch.uint8):
            raise ValueError(f"pcm dtype must be int16, float32 or uint8 (packed int24), got {pcm.dtype}")
        n = int(pcm.shape[0])
        ch = int(pcm.shape[1]) if pcm.dim() >= 2 else 1
        if ch == 1 and pcm.dtype == torch.float32 and int(sample_rate) == SAMPLE_RATE:
            return pcm.reshape(-1)                      # already what the model consumes
        n_out = int(self._lib.vtc_resampled_length(n, int(sample_rate)))
        if n_out < 0:
            raise ValueError(f"bad length / sample rate ({n}, {sample_rate})")
        out = torch.empty(max(n_out, 1), dtype=torch.float32, device=self.device)
        got = C.c_int64(0)
        fmt = 0 if pcm.dtype == torch.int16 else 1 if pcm.dtype == torch.float32 else 2
        L.check(self._lib.vtc_prepare_audio(self._handle, C.c_void_p(pcm.data_ptr()), fmt, n,
                                            ch, int(sample_rate), C.c_void_p(out.data_ptr()), n_out, C.byref(got),
                                            self._stream()), self._handle)
        return out[: got.value]

    # ------------------------------------------------------------------ decode
    def threshold_mask(self, logits: torch.Tensor, lo_logit, hi_logit) -> torch.Tensor:
        """(n_frames, n_labels) uint8 0/1 mask on the device (include/vtc_amd.h: vtc_threshold_mask)."""
        if logits.device != self.device or logits.dtype != torch.float32 or not logits.is_contiguous():
            raise ValueError("logits must be a contiguous float32 tensor on the engine's device")
        lo = np.ascontiguousarray(lo_logit, dtype=np.float32)
        hi = np.ascontiguousarray(hi_logit, dtype=np.float32)
        if len(lo) != self.n_labels or len(hi) != self.n_labels or logits.dim() != 2 or logits.shape[1] != self.n_labels:
            raise ValueError("need (n_frames, n_labels) logits and one lower and one upper bound per label")
        mask = torch.empty(logits.shape, dtype=torch.uint8, device=self.device)
        L.check(self._lib.vtc_threshold_mask(self._handle, C.c_void_p(logits.data_ptr()), int(logits.shape[0]),
                                             lo.ctypes.data_as(L._f32p), hi.ctypes.data_as(L._f32p),
                                             C.c_void_p(mask.data_ptr()), self._stream()), self._handle)
        return mask

    def decode(self, logits: torch.Tensor, lo_logit, hi_logit):
        """Threshold + run-length decode on the GPU (include/vtc_amd.h: vtc_decode).

        Returns (start_samples[int64], end_samples[int64], label_index[int32], max_logit, mean_sigmoid)."""
        if logits.device != self.device or logits.dtype != torch.float32 or not logits.is_contiguous():
            raise ValueError("logits must be a contiguous float32 tensor on the engine's device")
        n = int(logits.shape[0])
        lo = np.ascontiguousarray(lo_logit, dtype=np.float32)
        hi = np.ascontiguousarray(hi_logit, dtype=np.float32)
        if len(lo) != self.n_labels or len(hi) != self.n_labels:
            raise ValueError("need one lower and one upper bound per label")
        cap = max(1024, (n // 2 + 1) * self.n_labels)
        s = np.empty(cap, dtype=np.int64)
        e = np.empty(cap, dtype=np.int64)
        lab = np.empty(cap, dtype=np.int32)
        n_out, mx, mean = C.c_int64(0), C.c_float(0), C.c_double(0)
        L.check(self._lib.vtc_decode(self._handle, C.c_void_p(logits.data_ptr()), n, lo.ctypes.data_as(L._f32p),
                                     hi.ctypes.data_as(L._f32p), s.ctypes.data_as(L._i64p), e.ctypes.data_as(L._i64p),
                                     lab.ctypes.data_as(L._i32p), cap, C.byref(n_out), C.byref(mx), C.byref(mean),
                                     self._stream()), self._handle)
        k = n_out.value
        return s[:k].copy(), e[:k].copy(), lab[:k].copy(), float(mx.value), float(mean.value)

    # ------------------------------------------------------------------ test / bench hooks
    def debug_stop_after(self, stage: int) -> None:
        L.check(self._lib.vtc_debug_stop_after(self._handle, stage), self._handle)

    def debug_stage(self, stage: int, n_windows: int, n_elems: int) -> np.ndarray:
        out = np.empty(n_elems, dtype=np.float32)
        got = self._lib.vtc_debug_stage(self._handle, stage, n_windows, out.ctypes.data_as(L._f32p), n_elems)
        if got < 0:
            L.check(int(got), self._handle)
        return out[:got]

    def profile_enable(self, on: bool = True) -> None:
        L.check(self._lib.vtc_profile_enable(self._handle, 1 if on else 0), self._handle)

    def profile_read(self) -> dict:
        ms = (C.c_double * L.PROF_CLASSES)()
        cnt = (C.c_int64 * L.PROF_CLASSES)()
        fl = (C.c_double * L.PROF_CLASSES)()
        by = (C.c_double * L.PROF_CLASSES)()
        L.check(self._lib.vtc_profile_read(self._handle, ms, cnt, fl, by), self._handle)
        out = {}
        for c in range(L.PROF_CLASSES):
            name = self._lib.vtc_profile_class_name(c).decode()
            out[name] = dict(ms=ms[c], launches=int(cnt[c]), flops=fl[c], bytes=by[c])
        return out
