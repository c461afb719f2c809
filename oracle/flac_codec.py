"""TEST INFRASTRUCTURE ONLY -- never imported by the product (vtc_amd/).

FLAC members of the audio front-end.  The reference decodes FLAC through
torchcodec/ffmpeg inside segma (slurm/vtc.slurm:29,34; src/analysis/vtc_on_clips.py:163-184), neither of which is
installed here, so the oracle restates the published format: RFC 9639 ("Free Lossless Audio Codec", IETF, Dec 2024).

* `decode(data)`  -- scalar pure-Python decoder, one function per RFC section (9.1 frame header, 9.2 subframes,
                     9.2.5-9.2.7 predictors + Rice residual, 4.2 stereo decorrelation, 9.3 footer CRC).
* `encode(...)`   -- a small encoder (numpy) that produces the inputs of the parity tests; it can be steered into every
                     subframe type, both Rice parameter widths, escape partitions, wasted bits and all four stereo modes.

Pinned by: (1) the decoder reproduces the one-frame example stream of RFC 9639 appendix D.1 (bytes and expected
samples in tests/golden/flac_rfc9639_example1.json; the stream's own CRC-8, CRC-16 and MD5 all check, which also pins
the transcription of the bytes) and the second frame of the appendix D.2 stream (flac_rfc9639_example2_frame2.json,
self-checking through its CRCs); both are verbatim subframes with wasted bits; (2) lossless round trips encode -> decode == input with the MD5 in STREAMINFO
(RFC 8.2) compared against hashlib's MD5 of the input.  Streams written by libFLAC itself are not available offline:
beyond (1) the decoder is "parity unpinned" against real-encoder output.
"""

from __future__ import annotations

import hashlib

import numpy as np

# ----------------------------------------------------------------------------------------------- CRCs (RFC 9.1.8, 9.3)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


# ----------------------------------------------------------------------------------------------------------- decoder


class _Bits:
    def __init__(self, data: bytes, pos: int):
        self.d, self.pos = data, pos * 8

    def u(self, n: int) -> int:
        v = 0
        for _ in range(n):
            byte = self.d[self.pos >> 3]
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def s(self, n: int) -> int:
        v = self.u(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self) -> int:
        q = 0
        while self.u(1) == 0:
            q += 1
        return q

    def align(self):
        self.pos = (self.pos + 7) & ~7

    @property
    def byte(self) -> int:
        assert self.pos % 8 == 0
        return self.pos >> 3


_BLOCK = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608, 8: 256, 9: 512, 10: 1024, 11: 2048, 12: 4096, 13: 8192,
          14: 16384, 15: 32768}
_BPS = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}


def stream_info(data: bytes) -> dict:
    o = 0
    if data[:3] == b"ID3":
        sz = (data[6] & 0x7F) << 21 | (data[7] & 0x7F) << 14 | (data[8] & 0x7F) << 7 | (data[9] & 0x7F)
        o = 10 + sz + (10 if data[5] & 0x10 else 0)
    if data[o:o + 4] != b"fLaC":
        raise ValueError("not a FLAC stream")
    o += 4
    info = None
    while True:
        last, typ = data[o] >> 7, data[o] & 0x7F
        ln = int.from_bytes(data[o + 1:o + 4], "big")
        o += 4
        if typ == 0:
            s = data[o:o + 34]
            x = int.from_bytes(s[10:18], "big")
            info = {"min_block": int.from_bytes(s[0:2], "big"), "max_block": int.from_bytes(s[2:4], "big"),
                    "sample_rate": x >> 44, "channels": ((x >> 41) & 7) + 1, "bps": ((x >> 36) & 31) + 1,
                    "total": x & ((1 << 36) - 1), "md5": bytes(s[18:34])}
        o += ln
        if last:
            break
    if info is None:
        raise ValueError("no STREAMINFO")
    info["audio_offset"] = o
    return info


def _residual(br: _Bits, blocksize: int, order: int) -> list:
    method = br.u(2)
    if method > 1:
        raise ValueError("reserved residual coding method")
    pbits, esc = (5, 31) if method else (4, 15)
    porder = br.u(4)
    out = []
    for part in range(1 << porder):
        cnt = (blocksize >> porder) - (order if part == 0 else 0)
        k = br.u(pbits)
        if k == esc:
            nb = br.u(5)
            out.extend(br.s(nb) if nb else 0 for _ in range(cnt))
        else:
            for _ in range(cnt):
                u = (br.unary() << k) | br.u(k)
                out.append((u >> 1) ^ -(u & 1))
    return out


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(br: _Bits, blocksize: int, bps: int) -> list:
    if br.u(1):
        raise ValueError("subframe padding bit set")
    typ = br.u(6)
    wasted = br.unary() + 1 if br.u(1) else 0
    bps -= wasted
    if typ == 0:
        s = [br.s(bps)] * blocksize
    elif typ == 1:
        s = [br.s(bps) for _ in range(blocksize)]
    elif 8 <= typ <= 12 or typ >= 32:
        if typ >= 32:
            order = (typ & 31) + 1
            s = [br.s(bps) for _ in range(order)]
            prec = br.u(4) + 1
            shift = br.s(5)
            coef = [br.s(prec) for _ in range(order)]
        else:
            order = typ - 8
            s = [br.s(bps) for _ in range(order)]
            coef, shift = _FIXED[order], 0
        for r in _residual(br, blocksize, order):
            pred = sum(c * s[-1 - j] for j, c in enumerate(coef))
            s.append(r + (pred >> shift))
    else:
        raise ValueError("reserved subframe type")
    return [v << wasted for v in s]


def decode(data: bytes, check_md5: bool = True):
    """-> (int64 array (n, channels), info dict).  Verifies every frame's CRC-8/CRC-16 and (if present) the MD5."""
    info = stream_info(data)
    o, nch, chans = info["audio_offset"], info["channels"], [[] for _ in range(info["channels"])]
    done = 0
    while o < len(data) and not (info["total"] and done >= info["total"]):
        if data[o] != 0xFF or data[o + 1] & 0xFE != 0xF8:
            raise ValueError(f"lost sync at byte {o}")
        br = _Bits(data, o)
        br.u(16)
        bs_code, sr_code, ch_code, ss_code = br.u(4), br.u(4), br.u(4), br.u(3)
        br.u(1)
        b0 = br.u(8)
        extra = 0 if b0 < 0x80 else 1 if b0 < 0xE0 else 2 if b0 < 0xF0 else 3 if b0 < 0xF8 else 4 if b0 < 0xFC else 5 if b0 < 0xFE else 6
        for _ in range(extra):
            br.u(8)
        blocksize = br.u(8) + 1 if bs_code == 6 else br.u(16) + 1 if bs_code == 7 else _BLOCK[bs_code]
        if sr_code == 12:
            br.u(8)
        elif sr_code in (13, 14):
            br.u(16)
        if crc8(data[o:br.byte]) != br.u(8):
            raise ValueError(f"frame header CRC-8 mismatch at byte {o}")
        bps = info["bps"] if ss_code == 0 else _BPS[ss_code]
        n_frame_ch = ch_code + 1 if ch_code < 8 else 2
        if n_frame_ch != nch:
            raise ValueError("channel count changes")
        sub = []
        for c in range(n_frame_ch):
            side = (ch_code == 8 and c == 1) or (ch_code == 9 and c == 0) or (ch_code == 10 and c == 1)
            sub.append(_subframe(br, blocksize, bps + int(side)))
        br.align()
        if crc16(data[o:br.byte]) != br.u(16):
            raise ValueError(f"frame CRC-16 mismatch at byte {o}")
        o = br.byte
        if ch_code == 8:
            sub[1] = [l - s for l, s in zip(sub[0], sub[1])]
        elif ch_code == 9:
            sub[0] = [s + r for s, r in zip(sub[0], sub[1])]
        elif ch_code == 10:
            mids = [(m << 1) | (s & 1) for m, s in zip(sub[0], sub[1])]
            sub = [[(m + s) >> 1 for m, s in zip(mids, sub[1])], [(m - s) >> 1 for m, s in zip(mids, sub[1])]]
        take = blocksize if not info["total"] else min(blocksize, info["total"] - done)
        for c in range(nch):
            chans[c].extend(sub[c][:take])
        done += take
    pcm = np.array(chans, dtype=np.int64).T.reshape(done, nch)
    if check_md5 and any(info["md5"]):
        if pcm_md5(pcm, info["bps"]) != info["md5"]:
            raise ValueError("MD5 of the decoded samples differs from STREAMINFO")
    return pcm, info


def pcm_md5(pcm: np.ndarray, bps: int) -> bytes:
    """RFC 9639 8.2: MD5 of the interleaved samples, little-endian, each in ceil(bps/8) bytes."""
    nbytes = (bps + 7) // 8
    raw = np.ascontiguousarray(pcm.astype("<i8")).view(np.uint8).reshape(-1, 8)[:, :nbytes]
    return hashlib.md5(raw.tobytes()).digest()


# ----------------------------------------------------------------------------------------------------------- encoder


class _Writer:
    """Bit sink: fields are (value, nbits) runs collected as numpy bit arrays and packed once per frame."""

    def __init__(self):
        self.parts = []

    def u(self, v: int, n: int):
        if n:
            self.parts.append(np.array([(v >> (n - 1 - i)) & 1 for i in range(n)], dtype=np.uint8))

    def s(self, v: int, n: int):
        self.u(v & ((1 << n) - 1), n)

    def many_s(self, vals: np.ndarray, n: int):
        v = vals.astype(np.int64) & ((1 << n) - 1)
        shifts = np.arange(n - 1, -1, -1, dtype=np.int64)
        self.parts.append(((v[:, None] >> shifts) & 1).astype(np.uint8).ravel())

    def rice(self, res: np.ndarray, k: int):
        u = np.where(res >= 0, res << 1, ((-res) << 1) - 1).astype(np.int64)
        q = u >> k
        lens = q + 1 + k
        ends = np.cumsum(lens)
        bits = np.zeros(int(ends[-1]) if len(ends) else 0, dtype=np.uint8)
        stop = ends - k - 1
        bits[stop] = 1
        for b in range(k):
            bits[stop + 1 + b] = (u >> (k - 1 - b)) & 1
        self.parts.append(bits)

    def nbits(self) -> int:
        return int(sum(len(p) for p in self.parts))

    def align(self):
        r = (-self.nbits()) % 8
        if r:
            self.parts.append(np.zeros(r, dtype=np.uint8))

    def bytes(self) -> bytes:
        return np.packbits(np.concatenate(self.parts)).tobytes() if self.parts else b""


def _rice_cost(res: np.ndarray, k: int) -> int:
    u = np.where(res >= 0, res << 1, ((-res) << 1) - 1).astype(np.int64)
    return int(((u >> k) + 1 + k).sum())


def _write_residual(w: _Writer, res: np.ndarray, blocksize: int, order: int, porder: int, wide: bool, escape: bool):
    while porder > 0 and ((blocksize & ((1 << porder) - 1)) or (blocksize >> porder) <= order):
        porder -= 1
    pbits, esc = (5, 31) if wide else (4, 15)
    w.u(1 if wide else 0, 2)
    w.u(porder, 4)
    i = 0
    for part in range(1 << porder):
        cnt = (blocksize >> porder) - (order if part == 0 else 0)
        r = res[i:i + cnt]
        i += cnt
        if escape and part % 2 == 1:
            nb = max(1, int(np.abs(r).max(initial=0))).bit_length() + 1 if cnt and np.any(r) else 0
            w.u(esc, pbits)
            w.u(nb, 5)
            if nb:
                w.many_s(r, nb)
            continue
        kmax = min(esc - 1, 24)
        if cnt:
            mean_u = 2.0 * float(np.abs(r).mean())
            k0 = int(np.clip(np.floor(np.log2(mean_u)) if mean_u >= 1.0 else 0, 0, kmax))
            cands = sorted({max(k0 - 1, 0), k0, min(k0 + 1, kmax)})
            k = min(cands, key=lambda kk: _rice_cost(r, kk))
        else:
            k = 0
        w.u(k, pbits)
        if cnt:
            w.rice(r, k)


def _lpc_coefs(x: np.ndarray, order: int, prec: int):
    """Quantised forward predictor from the autocorrelation method (any coefficients give a valid stream)."""
    xf = x.astype(np.float64) * np.hanning(len(x))
    r = np.array([np.dot(xf[: len(xf) - l], xf[l:]) for l in range(order + 1)])
    if r[0] <= 0:
        return np.zeros(order, dtype=np.int64), 0
    R = np.array([[r[abs(i - j)] for j in range(order)] for i in range(order)]) + 1e-9 * r[0] * np.eye(order)
    a = np.linalg.solve(R, r[1:order + 1])
    amax = np.abs(a).max()
    shift = int(np.clip(prec - 1 - (int(np.floor(np.log2(amax))) + 1 if amax > 0 else 0), 0, 15))
    q = np.clip(np.round(a * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1).astype(np.int64)
    return q, shift


def _write_subframe(w: _Writer, x: np.ndarray, bps: int, mode: str, porder: int, wide: bool, escape: bool, lpc_order: int,
                    lpc_prec: int):
    n = len(x)
    wasted = 0
    if np.any(x):
        while not np.any(x & ((1 << (wasted + 1)) - 1)):
            wasted += 1
    else:
        mode = "constant"
    if wasted:
        x = x >> wasted
        bps -= wasted
    if mode != "verbatim" and np.all(x == x[0]):
        mode = "constant"
    if mode == "auto":
        best, best_cost = 0, None
        for o in range(0, min(4, n - 1) + 1):
            cost = int(np.abs(np.diff(x, o)).sum())
            if best_cost is None or cost < best_cost:
                best, best_cost = o, cost
        mode = f"fixed{best}"
    w.u(0, 1)
    if mode == "constant":
        w.u(0, 6)
    elif mode == "verbatim":
        w.u(1, 6)
    elif mode.startswith("fixed"):
        order = min(int(mode[5:]), n)
        w.u(8 + order, 6)
    else:
        order = min(lpc_order, n)
        w.u(32 + order - 1, 6)
    if wasted:
        w.u(1, 1)
        w.u(1, wasted)          # wasted-1 zeros then a one
    else:
        w.u(0, 1)
    if mode == "constant":
        w.s(int(x[0]), bps)
    elif mode == "verbatim":
        w.many_s(x, bps)
    elif mode.startswith("fixed"):
        res = np.diff(x, order)
        if order:
            w.many_s(x[:order], bps)
        _write_residual(w, res, n, order, porder, wide, escape)
    else:
        coef, shift = _lpc_coefs(x, order, lpc_prec)
        w.many_s(x[:order], bps)
        w.u(lpc_prec - 1, 4)
        w.s(shift, 5)
        for c in coef:
            w.s(int(c), lpc_prec)
        pred = np.zeros(n - order, dtype=np.int64)
        for j in range(order):
            pred += coef[j] * x[order - 1 - j: n - 1 - j]
        res = x[order:] - (pred >> shift)
        _write_residual(w, res, n, order, porder, wide, escape)


def _utf8_number(v: int) -> bytes:
    """RFC 9.1.5: the UTF-8 scheme extended to 36 bits (1..7 bytes)."""
    if v < 0x80:
        return bytes([v])
    nbytes = next(k for k, lim in ((2, 11), (3, 16), (4, 21), (5, 26), (6, 31), (7, 36)) if v < (1 << lim))
    tail = [0x80 | ((v >> (6 * i)) & 0x3F) for i in range(nbytes - 2, -1, -1)]
    lead = ((0xFF << (8 - nbytes)) & 0xFF) | (v >> (6 * (nbytes - 1)))
    return bytes([lead] + tail)


_SR_CODE = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10,
            96000: 11}
_BS_CODE = {v: k for k, v in _BLOCK.items()}
_SS_CODE = {v: k for k, v in _BPS.items()}


def encode(pcm: np.ndarray, sample_rate: int, bps: int = 16, blocksize: int = 4096, subframe: str = "auto",
           stereo: str = "auto", porder: int = 2, wide_rice: bool = False, escape: bool = False, lpc_order: int = 8,
           lpc_prec: int = 12, variable_blocks=None, with_total: bool = True, with_md5: bool = True,
           extra_metadata: bool = True) -> bytes:
    """pcm: integer array (n,) or (n, channels) with values in the bps-bit signed range -> FLAC stream.
    subframe: auto | constant | verbatim | fixed0..fixed4 | lpc;  stereo: auto | independent | left_side | side_right |
    mid_side;  variable_blocks: list of block sizes cycled through (variable-blocksize stream, sample-number coded)."""
    x = np.asarray(pcm).astype(np.int64)
    if x.ndim == 1:
        x = x[:, None]
    n, nch = x.shape
    frames = bytearray()
    pos = frame_no = 0
    min_bs, max_bs, min_fs, max_fs = 65535, 0, 1 << 24, 0
    while pos < n:
        bs = variable_blocks[frame_no % len(variable_blocks)] if variable_blocks else blocksize
        bs = min(bs, n - pos)
        blk = x[pos:pos + bs]
        hdr = _Writer()
        hdr.u(0xFFF8 | (1 if variable_blocks else 0), 16)
        bs_code = _BS_CODE.get(bs, 6 if bs <= 256 else 7)
        sr_code = _SR_CODE.get(sample_rate, 13 if sample_rate < 65536 else 0)
        hdr.u(bs_code, 4)
        hdr.u(sr_code, 4)
        mode = stereo if nch == 2 else "independent"
        if mode == "auto":
            l, r = blk[:, 0], blk[:, 1]
            cands = {"independent": (l, r), "left_side": (l, l - r), "side_right": (l - r, r),
                     "mid_side": ((l + r) >> 1, l - r)}
            mode = min(cands, key=lambda m: sum(int(np.abs(np.diff(c)).sum()) for c in cands[m]))
        ch_code = {"independent": nch - 1, "left_side": 8, "side_right": 9, "mid_side": 10}[mode]
        hdr.u(ch_code, 4)
        hdr.u(_SS_CODE.get(bps, 0), 3)
        hdr.u(0, 1)
        hb = hdr.bytes() + _utf8_number(pos if variable_blocks else frame_no)
        if bs_code == 6:
            hb += bytes([bs - 1])
        elif bs_code == 7:
            hb += (bs - 1).to_bytes(2, "big")
        if sr_code == 13:
            hb += sample_rate.to_bytes(2, "big")
        hb += bytes([crc8(hb)])
        w = _Writer()
        if mode == "independent":
            subs = [(blk[:, c], bps) for c in range(nch)]
        else:
            l, r = blk[:, 0], blk[:, 1]
            subs = {"left_side": [(l, bps), (l - r, bps + 1)], "side_right": [(l - r, bps + 1), (r, bps)],
                    "mid_side": [((l + r) >> 1, bps), (l - r, bps + 1)]}[mode]
        for sx, sb in subs:
            _write_subframe(w, np.ascontiguousarray(sx), sb, subframe, porder, wide_rice, escape, lpc_order, lpc_prec)
        w.align()
        body = hb + w.bytes()
        body += crc16(body).to_bytes(2, "big")
        frames += body
        min_bs, max_bs = min(min_bs, bs), max(max_bs, bs)
        min_fs, max_fs = min(min_fs, len(body)), max(max_fs, len(body))
        pos += bs
        frame_no += 1
    if n == 0:
        min_bs = max_bs = blocksize
        min_fs = max_fs = 0
    if not variable_blocks:
        min_bs = max_bs = blocksize       # RFC 8.2: min = max = the nominal block size for fixed-blocksize streams
    md5 = pcm_md5(x, bps) if with_md5 else bytes(16)
    packed = (sample_rate << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | (n if with_total else 0)
    si = (max(min_bs, 16).to_bytes(2, "big") + max(max_bs, 16).to_bytes(2, "big") + min_fs.to_bytes(3, "big") +
          max_fs.to_bytes(3, "big") + packed.to_bytes(8, "big") + md5)
    out = bytearray(b"fLaC")
    if extra_metadata:
        vendor = b"vtc_amd test encoder"
        vc = len(vendor).to_bytes(4, "little") + vendor + (0).to_bytes(4, "little")
        out += bytes([0]) + len(si).to_bytes(3, "big") + si
        out += bytes([4]) + len(vc).to_bytes(3, "big") + vc
        out += bytes([0x81]) + (16).to_bytes(3, "big") + bytes(16)      # padding block, last
    else:
        out += bytes([0x80]) + len(si).to_bytes(3, "big") + si
    return bytes(out + frames)
