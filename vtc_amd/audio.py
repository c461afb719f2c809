"""Audio front-end of the hot path: file -> mono 16 kHz float32 in [-1, 1).

The reference leaves decoding to segma (torchaudio/torchcodec, requirements.txt:393-394); the rule for
non-16 kHz or multi-channel input is unpinned.  Here: WAV (PCM 8/16/24/32, float, G.711 A-law /
mu-law, 4-bit IMA ADPCM, RF64 / BW64 headers; own RIFF parser or scipy), FLAC (the library's RFC 9639 decoder, vtc_flac_decode), AIFF / AIFF-C PCM; channels are averaged;
other rates are resampled with torchaudio's windowed-sinc polyphase filter (sinc_interp_hann, width 6, rolloff 0.99).
The lossy formats of scripts/make_manifest.py:26-39 (mp3, ogg, opus, m4a, aac, wma) need codecs that are not available
offline; they raise, which the pipeline turns into an error row.  An .ogg / .oga file is opened as far as its container: FLAC
in Ogg is decoded, Vorbis / Opus / Speex raise with the codec's name.

Two routes give the same samples: `read_pcm` + `VtcEngine.prepare_audio` (raw PCM is copied to the GPU and
mixed / resampled there: the pipeline's route) and `read_audio` (all on the host: tests, small tools).
"""

from __future__ import annotations

import os
from math import gcd
from pathlib import Path

import numpy as np

SAMPLE_RATE = 16_000


def to_float32(data: np.ndarray) -> np.ndarray:
    if data.dtype == np.int16:
        out = data.astype(np.float32) / np.float32(32768.0)
    elif data.dtype == np.int32:
        out = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        out = (data.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif data.dtype in (np.float32, np.float64):
        out = data.astype(np.float32, copy=False)
    else:
        raise ValueError(f"unsupported sample type {data.dtype}")
    return out


def pcm24_to_float32(packed: np.ndarray) -> np.ndarray:
    """Packed little-endian 24-bit PCM, uint8 (n, channels, 3) as read_pcm returns it -> float32 (n,) or (n, channels) =
    integer / 2^23 (exact in float32; what scipy's left-justified int32 / 2^31 gives).  Host route (read_audio, tools)."""
    b = np.asarray(packed).astype(np.int32)
    v = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
    v = np.where(v & 0x800000, v - (1 << 24), v)
    out = (v.astype(np.float32) / np.float32(8388608.0))
    return out[:, 0] if out.shape[1] == 1 else out


def g711_table(law: str) -> np.ndarray:
    """ITU-T G.711 expansion, 256 codes -> 16-bit linear PCM (the tables of the Recommendation, as every decoder has
    them).  mu-law: complement the byte; value = ((mantissa << 3) + 132 << exponent) - 132, sign bit set = negative
    (range +-32124).  A-law: XOR 0x55; exponent 0: (mantissa << 4) + 8, else ((mantissa << 4) + 264) << (exponent - 1);
    sign bit set = POSITIVE (range +-32256)."""
    code = np.arange(256, dtype=np.int32)
    if law == "ulaw":
        u = ~code & 0xFF
        mag = ((((u & 0x0F) << 3) + 0x84) << ((u >> 4) & 7)) - 0x84
        out = np.where(u & 0x80, -mag, mag)
    elif law == "alaw":
        a = code ^ 0x55
        e, m = (a >> 4) & 7, a & 0x0F
        mag = np.where(e == 0, (m << 4) + 8, ((m << 4) + 0x108) << np.maximum(e - 1, 0))
        out = np.where(a & 0x80, mag, -mag)
    else:
        raise ValueError(law)
    return out.astype(np.int16)


def _wav_layout(f):
    """Header of a RIFF/WAVE (or RF64 / BW64: 64-bit sizes in a 'ds64' chunk, for files past 4 GB) file open at position 0:
    (kind, channels, sample_rate, data offset, frames) with kind np.int16 / np.float32 for the two formats the GPU front-end
    takes as they are, "alaw" / "ulaw" for 8-bit G.711 (expanded to int16 by _fast_wav); None for anything else."""
    import struct

    head = f.read(12)
    if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64", b"BW64") or head[8:12] != b"WAVE":
        return None
    fmt, data64 = None, None
    while True:
        hdr = f.read(8)
        if len(hdr) < 8:
            return None
        cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
        if cid == b"fmt ":
            body = f.read(size + (size & 1))
            if size < 16:
                return None
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and size >= 40:                       # WAVE_FORMAT_EXTENSIBLE: sub-format GUID starts with the tag
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"ds64":
            body = f.read(size + (size & 1))
            if size >= 16:
                data64 = struct.unpack("<Q", body[8:16])[0]
        elif cid == b"data":
            if fmt is None:
                return None
            tag, ch, sr, bits = fmt
            kind = (np.int16 if (tag == 1 and bits == 16) else np.float32 if (tag == 3 and bits == 32) else
                    "alaw" if (tag == 6 and bits == 8) else "ulaw" if (tag == 7 and bits == 8) else
                    "pcm24" if (tag == 1 and bits == 24) else None)
            if kind is None or ch < 1:
                return None
            width = 3 if kind == "pcm24" else 1 if isinstance(kind, str) else np.dtype(kind).itemsize
            offset = f.tell()
            avail = os.fstat(f.fileno()).st_size - offset
            if size == 0xFFFFFFFF and data64 is not None:
                size = data64
            nbytes = min(size, avail) if size not in (0, 0xFFFFFFFF) else avail
            return kind, ch, int(sr), offset, nbytes // (ch * width)
        else:
            f.seek(size + (size & 1), 1)


def _fast_wav(path: Path):
    """Minimal RIFF/WAVE reader for the formats the GPU front-end takes as they are (PCM16, float32; G.711 A-law / mu-law
    is expanded to int16 first): one header parse and one read (small files) or a memory map (large ones).  Returns None
    for anything else - scipy handles it."""
    with open(path, "rb") as f:
        lay = _wav_layout(f)
        if lay is None:
            return None
        kind, ch, sr, offset, n = lay
        dtype = np.uint8 if isinstance(kind, str) else kind
        count = n * ch * (3 if kind == "pcm24" else 1)
        if count * np.dtype(dtype).itemsize >= (8 << 20):
            data = np.memmap(path, dtype=dtype, mode="r", offset=offset, shape=(count,))
        else:
            f.seek(offset)
            data = np.fromfile(f, dtype=dtype, count=count)
        if kind == "pcm24":
            return data.reshape(n, ch, 3), sr      # packed bytes as stored: the GPU front-end unpacks them (format 2)
        if isinstance(kind, str):
            data = g711_table(kind)[data]          # 8-bit companded -> int16: from here on the PCM16 route
        return (data.reshape(n, ch) if ch > 1 else data), sr


def read_plain_wav_into(path, dst: np.ndarray, pos: int):
    """Bundle staging of short files (pipeline/vtc.py): if `path` is a mono 16 kHz PCM16 WAV whose samples fit dst[pos:],
    read them straight into that int16 array (pinned staging memory) and return their count; None for any other file
    (other formats, rates, channel counts, or no room), which the caller then stages on its own."""
    if Path(path).suffix.lower() not in (".wav", ".wave"):
        return None
    with open(path, "rb", buffering=0) as f:
        lay = _wav_layout(f)
        if lay is None:
            return None
        dtype, ch, sr, offset, n = lay
        if dtype is not np.int16 or ch != 1 or sr != SAMPLE_RATE or pos + n > dst.shape[0]:
            return None
        f.seek(offset)
        got = f.readinto(memoryview(dst[pos:pos + n]).cast("B")) if n else 0
        return got // 2


FLAC_THREADS = max(1, int(os.environ.get("VTC_FLAC_THREADS", "4")))      # threads per stream of 32 MB or more


def decode_flac(data, verify_md5: bool = False, n_threads: int | None = None):
    """FLAC stream held in memory (bytes, bytearray, or a uint8 array / memmap) -> (samples, sample_rate); samples are
    int16 as stored for 16-bit streams (the common case: half the PCIe bytes of float), else float32 = integer /
    2^(bits-1); shape (n,) or (n, channels).  Frame CRCs are always checked; verify_md5 also checks STREAMINFO's MD5
    of the decoded integers.  Runs in libvtc_amd.so outside the GIL."""
    import ctypes as C

    from . import lib as L

    lib = L.load()
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
    ptr, nbytes = C.c_void_p(buf.ctypes.data), C.c_int64(buf.size)
    sr, ch, bps, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    md5 = (C.c_uint8 * 16)()
    L.check(lib.vtc_flac_info(ptr, nbytes, C.byref(sr), C.byref(ch), C.byref(bps), C.byref(total), md5))
    n = total.value
    got = C.c_int64()
    if n == 0:                                       # length not declared: count first
        L.check(lib.vtc_flac_decode(ptr, nbytes, None, 2, 0, C.byref(got)))
        n = got.value
    want_md5 = verify_md5 and any(md5)
    if bps.value == 16:
        fmt, dtype = 0, np.int16
    elif want_md5:
        fmt, dtype = 2, np.int32
    else:
        fmt, dtype = 1, np.float32
    out = np.empty((n, ch.value), dtype=dtype)
    if n_threads is None:
        n_threads = FLAC_THREADS if buf.size >= (32 << 20) else 1
    L.check(lib.vtc_flac_decode_threads(ptr, nbytes, C.c_void_p(out.ctypes.data), fmt, n, C.byref(got), int(n_threads)))
    out = out[:got.value]
    if want_md5:
        import hashlib

        width = (bps.value + 7) // 8
        raw = out.tobytes() if width == out.itemsize else \
            np.ascontiguousarray(out.astype("<i4")).view(np.uint8).reshape(-1, 4)[:, :width].tobytes()
        if hashlib.md5(raw).digest() != bytes(md5):
            raise ValueError("FLAC: MD5 of the decoded samples differs from the stream's STREAMINFO")
        if fmt == 2:
            out = (out.astype(np.float64) / float(1 << (bps.value - 1))).astype(np.float32)
    return (out[:, 0] if ch.value == 1 else out), int(sr.value)


def decode_ima_adpcm_wav(data):
    """RIFF/WAVE with 4-bit IMA / DVI ADPCM (format tag 0x0011) held in memory -> (int16 samples (n,) or (n, channels),
    sample_rate), or None when the file is something else.  The blocks are decoded by libvtc_amd.so (vtc_ima_adpcm_decode,
    outside the GIL); the 'fact' chunk, when present, gives the true length (the last block is padded)."""
    import ctypes as C
    import struct

    from . import lib as L

    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8)
    if buf.size < 12 or bytes(buf[:4]) != b"RIFF" or bytes(buf[8:12]) != b"WAVE":
        return None
    pos, fmt, fact, chunk = 12, None, None, None
    while pos + 8 <= buf.size:
        cid, size = bytes(buf[pos:pos + 4]), struct.unpack("<I", bytes(buf[pos + 4:pos + 8]))[0]
        body = pos + 8
        if cid == b"fmt " and size >= 16:
            tag, ch, sr, _, align, bits = struct.unpack("<HHIIHH", bytes(buf[body:body + 16]))
            spb = struct.unpack("<H", bytes(buf[body + 18:body + 20]))[0] if size >= 20 else 0
            fmt = (tag, ch, sr, align, bits, spb)
        elif cid == b"fact" and size >= 4:
            fact = struct.unpack("<I", bytes(buf[body:body + 4]))[0]
        elif cid == b"data":
            chunk = (body, min(size, buf.size - body) if size not in (0, 0xFFFFFFFF) else buf.size - body)
            break
        pos = body + size + (size & 1)
    if fmt is None or chunk is None or fmt[0] != 0x11:
        return None
    tag, ch, sr, align, bits, spb = fmt
    if bits != 4 or ch < 1 or align < 4 * ch:
        raise ValueError(f"IMA ADPCM WAV: unsupported layout ({bits} bits per sample, {ch} channels, blocks of {align} bytes)")
    full = 1 + (align - 4 * ch) // (4 * ch) * 8
    cap = -(-chunk[1] // align) * (spb or full)
    if fact is not None and 0 < fact < cap:
        cap = fact
    out = np.empty((cap, ch), dtype=np.int16)
    got = C.c_int64()
    L.check(L.load().vtc_ima_adpcm_decode(C.c_void_p(buf.ctypes.data + chunk[0]), C.c_int64(chunk[1]), ch, align, spb,
                                          C.c_void_p(out.ctypes.data), cap, C.byref(got)))
    out = out[:got.value]
    return (out[:, 0] if ch == 1 else out), int(sr)


def decode_aiff(data):
    """AIFF / AIFF-C with uncompressed PCM ('NONE', 'twos', 'sowt', 'fl32'/'FL32') or G.711 ('ulaw', 'alaw') ->
    (samples, sample_rate)."""
    import struct

    buf = memoryview(data)
    if len(buf) < 12 or bytes(buf[:4]) != b"FORM" or bytes(buf[8:12]) not in (b"AIFF", b"AIFC"):
        raise ValueError("not an AIFF file")
    aifc = bytes(buf[8:12]) == b"AIFC"
    o, comm, ssnd = 12, None, None
    while o + 8 <= len(buf):
        cid, size = bytes(buf[o:o + 4]), struct.unpack(">I", buf[o + 4:o + 8])[0]
        body = o + 8
        if cid == b"COMM":
            ch, nframes, bits = struct.unpack(">hIh", buf[body:body + 8])
            expo, mant = struct.unpack(">HQ", buf[body + 8:body + 18])      # 80-bit extended sample rate
            rate = (mant / float(1 << 63)) * 2.0 ** ((expo & 0x7FFF) - 16383) if mant else 0.0
            ctype = bytes(buf[body + 18:body + 22]) if aifc and size >= 22 else b"NONE"
            comm = (ch, nframes, bits, int(round(rate)), ctype)
        elif cid == b"SSND":
            offset = struct.unpack(">I", buf[body:body + 4])[0]
            ssnd = (body + 8 + offset, min(size - 8 - offset, len(buf) - body - 8 - offset))
        o = body + size + (size & 1)
    if comm is None or ssnd is None:
        raise ValueError("AIFF file without COMM or SSND chunk")
    ch, nframes, bits, rate, ctype = comm
    start, nbytes = ssnd
    g711 = {b"ulaw": "ulaw", b"ULAW": "ulaw", b"alaw": "alaw", b"ALAW": "alaw"}.get(ctype)
    width = 1 if g711 else (bits + 7) // 8          # AIFF-C G.711: one byte per sample whatever COMM's sampleSize says (16)
    n = min(nframes, nbytes // (width * ch)) if ch > 0 and width > 0 else 0
    raw = np.frombuffer(buf, dtype=np.uint8, count=n * ch * width, offset=start)
    if g711:
        x = g711_table(g711)[raw]
    elif ctype in (b"fl32", b"FL32"):
        x = raw.view(">f4").astype(np.float32)
    elif ctype in (b"NONE", b"twos", b"sowt"):
        order = "<" if ctype == b"sowt" else ">"
        if width == 2:
            x = raw.view(order + "i2").astype(np.int16)
        elif width == 1:
            x = raw.view(np.int8).astype(np.float32) / np.float32(128.0)
        elif width == 4:
            x = (raw.view(order + "i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        elif width == 3:
            # packed 24-bit: handed on as little-endian bytes, uint8 (n, channels, 3), which the GPU front-end unpacks
            # (vtc_prepare_audio format 2; read_audio: pcm24_to_float32); big-endian files cost one byte-reversing copy
            p = raw.reshape(n, ch, 3)
            return (p if order == "<" else np.ascontiguousarray(p[..., ::-1])), rate
        else:
            raise ValueError(f"AIFF: unsupported sample size {bits}")
    else:
        raise ValueError(f"AIFF-C compression {ctype!r} is not supported")
    return (x.reshape(n, ch) if ch > 1 else x), rate


def _file_bytes(path: Path):
    size = path.stat().st_size
    return np.memmap(path, dtype=np.uint8, mode="r") if size >= (8 << 20) else np.fromfile(path, dtype=np.uint8)


UNSUPPORTED_LOSSY = (".mp3", ".ogg", ".opus", ".m4a", ".aac", ".wma")

_OGG_CRC = None


def ogg_crc32(page) -> int:
    """CRC of an Ogg page (RFC 3533 section 6: polynomial 0x04c11db7, initial value 0, no bit reflection, no final XOR;
    computed with the page's own checksum field zeroed).  Not zlib's CRC-32."""
    global _OGG_CRC
    if _OGG_CRC is None:
        t = []
        for i in range(256):
            r = i << 24
            for _ in range(8):
                r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
            t.append(r)
        _OGG_CRC = t
    t, crc = _OGG_CRC, 0
    for b in bytes(page):
        crc = ((crc << 8) & 0xFFFFFFFF) ^ t[((crc >> 24) ^ b) & 0xFF]
    return crc


OGG_CODECS = ((b"\x7fFLAC", "FLAC"), (b"\x01vorbis", "Vorbis"), (b"OpusHead", "Opus"), (b"Speex   ", "Speex"), (b"\x80theora", "Theora"))


def ogg_packets(data, check_pages: int = 8):
    """Packets of the first logical bitstream of an Ogg file (RFC 3533): pages are 'OggS' + a 27-byte header + a segment table;
    a packet is the run of segments up to the first one shorter than 255 bytes, possibly continued on the next page.  The page
    CRC is verified on the first `check_pages` pages (the header packets; -1 = all): audio packets of the one codec decoded here
    (FLAC) carry their own frame CRCs, which the FLAC decoder checks.  Returns a list of memoryviews / bytes."""
    import struct

    buf = memoryview(np.ascontiguousarray(data).view(np.uint8)) if isinstance(data, np.ndarray) else memoryview(data)
    pos, n, serial, page_no = 0, len(buf), None, 0
    packets, parts = [], []
    while pos + 27 <= n:
        if bytes(buf[pos:pos + 4]) != b"OggS":
            raise ValueError(f"Ogg: no page at byte {pos}")
        version, flags, _gran, ser, _seq, crc, nseg = struct.unpack("<BBqIIIB", buf[pos + 4:pos + 27])
        if version != 0 or pos + 27 + nseg > n:
            raise ValueError("Ogg: bad page header")
        table = bytes(buf[pos + 27:pos + 27 + nseg])
        body = pos + 27 + nseg
        size = sum(table)
        if body + size > n:
            raise ValueError("Ogg: truncated page")
        if check_pages < 0 or page_no < check_pages:
            page = bytearray(buf[pos:body + size])
            page[22:26] = b"\0\0\0\0"
            if ogg_crc32(page) != crc:
                raise ValueError(f"Ogg: page {page_no} fails its CRC")
        if serial is None:
            serial = ser
        if ser == serial:                       # other logical streams of a multiplexed file are skipped
            if not (flags & 1) and parts:       # a fresh packet starts although the previous one never ended: drop the torso
                parts = []
            at = body
            for seg in table:
                parts.append(buf[at:at + seg])
                at += seg
                if seg < 255:
                    packets.append(parts[0] if len(parts) == 1 else b"".join(parts))
                    parts = []
        pos = body + size
        page_no += 1
    if not packets:
        raise ValueError("Ogg: no packets")
    return packets


def decode_ogg(data):
    """Ogg file held in memory -> (samples, sample_rate).  FLAC in Ogg (the mapping of RFC 9639 section 10.1: a first packet
    0x7F 'FLAC' major minor n_header_packets 'fLaC' STREAMINFO, then one metadata block per header packet, then one FLAC frame
    per packet) is re-assembled into a native FLAC stream and decoded by the library (frame CRCs checked).  Vorbis, Opus and
    Speex streams raise with the codec's name: there is no decoder for them in this build."""
    packets = ogg_packets(data)
    head = bytes(packets[0][:16])
    codec = next((name for magic, name in OGG_CODECS if head.startswith(magic)), None)
    if codec != "FLAC":
        raise ValueError(f"Ogg stream carries {codec or 'an unknown codec'}; this build decodes FLAC in Ogg only "
                         "(no Vorbis / Opus decoder is available offline)")
    first = bytes(packets[0])
    if len(first) < 13 + 4 + 34 or first[9:13] != b"fLaC" or first[5] != 1:
        raise ValueError("Ogg FLAC: malformed identification packet (mapping version 1.x expected)")
    n_headers = (first[7] << 8) | first[8]      # 0 = unknown: metadata packets are then told from frames by their first bytes
    meta, k = [first[13:]], 1
    while k < len(packets) and (k <= n_headers if n_headers else bytes(packets[k][:1]) != b"\xff"):
        meta.append(bytes(packets[k]))
        k += 1
    # exactly one block may carry the last-metadata-block flag: the last one
    blocks = [bytes([b[0] & 0x7F]) + b[1:] for b in meta[:-1]] + [bytes([meta[-1][0] | 0x80]) + meta[-1][1:]]
    stream = b"fLaC" + b"".join(blocks) + b"".join(bytes(p) if not isinstance(p, bytes) else p for p in packets[k:])
    return decode_flac(stream)


def read_pcm(path):
    """Decode only as far as the container: (samples, sample_rate) with samples int16 or float32, shape (n,) or
    (n, channels), memory-mapped when large; 24-bit PCM WAVs come back as their packed bytes, uint8 (n, channels, 3)
    (VtcEngine.prepare_audio unpacks them on the GPU, pcm24_to_float32 on the host).  Mixing and resampling are left to the GPU."""
    path = Path(path)
    ext = path.suffix.lower()
    if ext == ".flac":
        return decode_flac(_file_bytes(path))
    if ext in (".aiff", ".aif", ".aifc"):
        return decode_aiff(_file_bytes(path))
    if ext in (".ogg", ".oga", ".opus"):
        try:
            return decode_ogg(_file_bytes(path))
        except ValueError as e:
            raise ValueError(f"{path.name}: {e}") from None
    if ext not in (".wav", ".wave"):
        why = "needs a codec that is not available in this build" if ext in UNSUPPORTED_LOSSY else "unknown audio format"
        raise ValueError(f"{path.name}: '{ext}' {why} (supported: wav, flac, aiff)")
    fast = _fast_wav(path)
    if fast is not None:
        return fast
    adpcm = decode_ima_adpcm_wav(_file_bytes(path))
    if adpcm is not None:
        return adpcm
    from scipy.io import wavfile

    try:
        sr, data = wavfile.read(str(path), mmap=True)
    except ValueError:
        sr, data = wavfile.read(str(path), mmap=False)   # 24-bit PCM cannot be memory-mapped
    if data.dtype not in (np.int16, np.float32):
        data = to_float32(np.asarray(data))
    return data, int(sr)


def sinc_resample_kernel(sr_in: int, sr_out: int = SAMPLE_RATE):
    """torchaudio.functional.resample's filter (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99):
    (orig, new, width, kernels float32 [new, 2*width+orig]), computed in float64."""
    g = gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    lpw, rolloff = 6.0, 0.99
    base_freq = min(orig, new) * rolloff
    width = int(np.ceil(lpw * orig / base_freq))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base_freq
    t = np.clip(t, -lpw, lpw)
    window = np.cos(t * np.pi / lpw / 2.0) ** 2
    t = t * np.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0.0, 1.0, np.sin(t) / t) * window * (base_freq / orig)
    return orig, new, width, k.astype(np.float32)


def sinc_resample(x: np.ndarray, sr_in: int) -> np.ndarray:
    """Host form of the GPU resampler (vtc_prepare_audio): float32 mono at sr_in -> 16 kHz."""
    orig, new, width, k = sinc_resample_kernel(sr_in)
    x = np.asarray(x, dtype=np.float32)
    if orig == new:
        return x
    n = x.shape[0]
    n_out = -(-new * n // orig)
    blocks = n // orig + 1
    pad = np.zeros(width + blocks * orig + width + orig, dtype=np.float32)
    pad[width:width + n] = x
    win = np.lib.stride_tricks.sliding_window_view(pad, k.shape[1])[::orig][:blocks]      # [blocks, K]
    out = (win.astype(np.float32) @ k.T.astype(np.float32)).reshape(-1)                   # block-major, phase-minor
    return np.ascontiguousarray(out[:n_out], dtype=np.float32)


def read_audio(path) -> np.ndarray:
    """Decode one file to mono 16 kHz float32.  Raises on anything unreadable; the host turns the
    exception into an error row (reference src/pipeline/vtc.py:245-249)."""
    path = Path(path)
    data, sr = read_pcm(path)
    x = pcm24_to_float32(data) if (data.dtype == np.uint8 and data.ndim == 3) else to_float32(np.asarray(data))
    if x.ndim == 2:
        x = x.mean(axis=1, dtype=np.float32)
    if sr != SAMPLE_RATE:
        x = sinc_resample(x, int(sr))
    return np.ascontiguousarray(x, dtype=np.float32)


def write_wav(path, samples: np.ndarray, sample_rate: int = SAMPLE_RATE) -> None:
    """Write float audio as 16-bit PCM (the format of the synthetic benchmark files)."""
    from scipy.io import wavfile

    pcm = np.clip(np.round(np.asarray(samples, dtype=np.float64) * 32768.0), -32768, 32767).astype(np.int16)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    wavfile.write(str(path), sample_rate, pcm)
