"""FLAC / AIFF members of the audio front-end: the library's host-side decoder against the
RFC 9639 example stream, against the pure-Python oracle decoder, and through lossless round trips with the test encoder
of oracle/flac_codec.py (every subframe type, Rice width, escape partition, wasted bits, stereo mode, sample size).
CPU only: vtc_flac_* do no device work."""

import ctypes as C
import json
import struct
from pathlib import Path

import numpy as np
import pytest

from oracle import flac_codec as F
from vtc_amd import lib as L
from vtc_amd.audio import decode_aiff, decode_flac, read_audio, read_pcm, write_wav

GOLDEN = Path(__file__).parent / "golden"


def tone(n, nch, bps, seed=0):
    rng = np.random.default_rng(seed)
    amp = (1 << (bps - 1)) * 0.4
    t = np.arange(n)
    x = amp * np.sin(2 * np.pi * t * 220 / 16000)[:, None] * np.array([1.0, 0.7])[:nch] + rng.normal(0, amp * 0.01, (n, nch))
    return x.astype(np.int64)


def as_decoded(x, bps):
    """What decode_flac returns for integer input x: int16 as stored for 16-bit streams, else integer / 2^(bps-1)."""
    y = x.astype(np.int16) if bps == 16 else (x / float(1 << (bps - 1))).astype(np.float32)
    return y[:, 0] if y.shape[1] == 1 else y


def test_rfc9639_example_stream():
    g = json.loads((GOLDEN / "flac_rfc9639_example1.json").read_text())
    data = bytes.fromhex(g["stream_hex"])
    want = np.array(g["samples"], dtype=np.int64)
    pcm, info = F.decode(data)                       # oracle: CRC-8, CRC-16 and MD5 of the stream all verified inside
    assert np.array_equal(pcm, want)
    assert (info["sample_rate"], info["channels"], info["bps"], info["total"]) == (44100, 2, 16, 1)
    got, sr = decode_flac(data, verify_md5=True)     # product decoder
    assert sr == g["sample_rate"] and got.dtype == np.int16 and np.array_equal(got, want)


def test_rfc9639_second_example_frame():
    g = json.loads((GOLDEN / "flac_rfc9639_example2_frame2.json").read_text())
    packed = (g["sample_rate"] << 44) | ((g["channels"] - 1) << 41) | ((g["bits_per_sample"] - 1) << 36) | len(g["samples"])
    si = (16).to_bytes(2, "big") * 2 + bytes(3) + (23).to_bytes(3, "big") + packed.to_bytes(8, "big") + bytes(16)
    data = b"fLaC" + bytes([0x80, 0, 0, 0x22]) + si + bytes.fromhex(g["frame_hex"])
    want = np.array(g["samples"], dtype=np.int64)
    assert want[:, 1].tolist() == [-9072, -8958, -9410] and not (want[:, 1] & 1).any()      # one wasted bit
    pcm, _ = F.decode(data)
    got, sr = decode_flac(data)
    assert np.array_equal(pcm, want) and np.array_equal(got, want) and sr == 44100


@pytest.mark.parametrize("bps", [8, 12, 16, 20, 24])
@pytest.mark.parametrize("nch", [1, 2])
def test_lossless_round_trip_all_subframe_types(bps, nch):
    x = tone(9000, nch, bps, seed=bps * 2 + nch)
    stereo_modes = ("auto", "independent", "left_side", "side_right", "mid_side") if nch == 2 else ("independent",)
    for sub in ("auto", "verbatim", "fixed0", "fixed1", "fixed2", "fixed3", "fixed4", "lpc", "constant"):
        for st in stereo_modes:
            for wide, esc in ((False, False), (True, True)):
                xx = np.full_like(x, -37) if sub == "constant" else x
                if sub == "fixed2":
                    xx = (xx >> 3) << 3                                   # wasted bits
                enc = F.encode(xx, 16000, bps=bps, blocksize=4096 if not wide else 1000,
                               subframe="auto" if sub == "constant" else sub, stereo=st, porder=3, wide_rice=wide,
                               escape=esc, with_total=not esc)
                got, sr = decode_flac(enc, verify_md5=True)
                want = as_decoded(xx, bps)
                assert sr == 16000
                assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), (sub, st, wide)


def test_product_decoder_matches_oracle_decoder():
    x = tone(2500, 2, 16, seed=3)
    for kw in (dict(subframe="lpc", stereo="mid_side", blocksize=576, lpc_order=12, lpc_prec=14),
               dict(subframe="auto", stereo="left_side", blocksize=192, wide_rice=True, escape=True),
               dict(subframe="fixed4", stereo="side_right", variable_blocks=[576, 192, 1000, 17]),
               dict(subframe="lpc", stereo="independent", blocksize=300, lpc_order=32, lpc_prec=15, with_total=False),
               dict(subframe="verbatim", stereo="auto", blocksize=4096, extra_metadata=False)):
        enc = F.encode(x, 44100, **kw)
        ref, info = F.decode(enc)
        got, sr = decode_flac(enc, verify_md5=True)
        assert np.array_equal(ref, x) and np.array_equal(got, ref.astype(np.int16)) and sr == 44100


def test_extreme_values_and_32_bit():
    for bps in (16, 24, 32):
        lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
        rng = np.random.default_rng(bps)
        x = rng.integers(lo, hi + 1, size=(3000, 2), dtype=np.int64)
        x[:4] = [[lo, hi], [hi, lo], [lo, lo], [hi, hi]]
        for st in ("independent", "mid_side", "left_side", "side_right"):          # the side channel needs bps+1 bits
            enc = F.encode(x, 48000, bps=bps, subframe="verbatim" if st != "independent" else "fixed1", stereo=st,
                           wide_rice=True, blocksize=1024)
            ref, _ = F.decode(enc[:])
            assert np.array_equal(ref, x)
            n, got = C.c_int64(), np.empty((3000, 2), dtype=np.int32)
            L.check(L.load().vtc_flac_decode(enc, len(enc), C.c_void_p(got.ctypes.data), 2, 3000, C.byref(n)))
            assert n.value == 3000 and np.array_equal(got, x)


def test_id3_prefix_unknown_length_and_trailing_bytes():
    x = tone(5000, 1, 16)
    enc = F.encode(x, 16000, with_total=False)
    tag = b"ID3\x04\x00\x00" + bytes([0, 0, 0, 20]) + bytes(20)
    got, _ = decode_flac(tag + enc)
    assert np.array_equal(got, x[:, 0])
    enc2 = F.encode(x, 16000) + b"TAG" + bytes(125)           # ID3v1 tag after the last frame
    got2, _ = decode_flac(enc2, verify_md5=True)
    assert np.array_equal(got2, x[:, 0])


def test_damaged_streams_raise():
    x = tone(9000, 1, 16)
    enc = bytearray(F.encode(x, 16000))
    with pytest.raises(L.VtcError, match="fLaC"):
        decode_flac(b"RIFF" + bytes(enc[4:]))
    bad = bytearray(enc)
    bad[len(bad) // 2] ^= 0x10                                 # one flipped bit in the middle of a frame
    with pytest.raises(L.VtcError, match="CRC|sync|subframe|Rice|residual|reserved"):
        decode_flac(bytes(bad))
    with pytest.raises(L.VtcError, match="ends after|truncated|CRC|sync"):
        decode_flac(bytes(enc[: len(enc) * 2 // 3]))
    with pytest.raises(L.VtcError, match="metadata|STREAMINFO"):
        decode_flac(bytes(enc[:20]))
    hdr = bytearray(enc)
    hdr[F.stream_info(bytes(enc))["audio_offset"] + 2] ^= 0x01  # frame header byte: CRC-8 must catch it
    with pytest.raises(L.VtcError, match="header CRC|reserved|invalid"):
        decode_flac(bytes(hdr))
    wrong = bytearray(enc)
    wrong[4 + 4 + 18] ^= 0xFF                                  # STREAMINFO MD5
    with pytest.raises(ValueError, match="MD5"):
        decode_flac(bytes(wrong), verify_md5=True)
    n = C.c_int64()
    out = np.empty(100, dtype=np.int16)
    rc = L.load().vtc_flac_decode(bytes(enc), len(enc), C.c_void_p(out.ctypes.data), 0, 100, C.byref(n))
    assert rc == L.ERR_CAPACITY
    enc24 = F.encode(tone(100, 1, 24), 16000, bps=24)
    rc = L.load().vtc_flac_decode(enc24, len(enc24), C.c_void_p(out.ctypes.data), 0, 100, C.byref(n))
    assert rc == L.ERR_ARG and "16 bits" in L.last_error(None)


def test_flac_file_reads_like_the_wav_of_the_same_samples(tmp_path):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((44100 * 2, 2)) * 3000).astype(np.int16)
    from scipy.io import wavfile

    wavfile.write(tmp_path / "a.wav", 44100, x)
    (tmp_path / "a.flac").write_bytes(F.encode(x.astype(np.int64), 44100, subframe="lpc"))
    pw, srw = read_pcm(tmp_path / "a.wav")
    pf, srf = read_pcm(tmp_path / "a.flac")
    assert srw == srf == 44100 and pf.dtype == np.int16 and np.array_equal(np.asarray(pw), pf)
    assert np.array_equal(read_audio(tmp_path / "a.wav"), read_audio(tmp_path / "a.flac"))     # mono mix + resample too
    with pytest.raises(ValueError, match="codec"):
        read_pcm(tmp_path / "a.mp3")


def test_aiff_against_the_standard_library_writer(tmp_path):
    import aifc                                                 # stdlib (deprecated, present in 3.10): independent writer
    import warnings

    rng = np.random.default_rng(6)
    for width, nch, rate in ((2, 2, 44100), (2, 1, 16000), (3, 1, 22050), (1, 2, 8000), (4, 1, 48000)):
        bits = width * 8
        x = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(4000, nch), dtype=np.int64)
        raw = x.astype(">i8").view(np.uint8).reshape(-1, 8)[:, 8 - width:].tobytes()
        path = tmp_path / f"a{width}{nch}.aiff"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with aifc.open(str(path), "wb") as f:
                f.setnchannels(nch)
                f.setsampwidth(width)
                f.setframerate(rate)
                f.writeframes(raw)
        got, sr = read_pcm(path)
        want = x.astype(np.int16) if width == 2 else (x / float(1 << (bits - 1))).astype(np.float32)
        want = want[:, 0] if nch == 1 else want
        if width == 3:      # packed little-endian bytes for the GPU front-end; the host route unpacks them exactly
            from vtc_amd.audio import pcm24_to_float32
            assert got.dtype == np.uint8 and got.shape == (4000, nch, 3)
            got = pcm24_to_float32(got)
        assert sr == rate and got.dtype == want.dtype and np.array_equal(got, want), (width, nch)
    # AIFF-C little-endian PCM ('sowt'), written by hand
    x = rng.integers(-32768, 32768, size=(1000, 2), dtype=np.int64).astype(np.int16)
    comm = struct.pack(">hIh", 2, 1000, 16) + struct.pack(">HQ", 16383 + 15, 48000 << 48) + b"sowt" + b"\x00\x00"
    ssnd = struct.pack(">II", 0, 0) + x.astype("<i2").tobytes()
    body = b"AIFC" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    got, sr = decode_aiff(b"FORM" + struct.pack(">I", len(body)) + body)
    assert sr == 48000 and np.array_equal(got, x)


def test_decoder_is_thread_safe_and_errors_are_per_thread():
    """vtc_flac_decode runs on the pipeline's reader threads at the same time (ctypes releases the GIL): many threads,
    different streams, every result exact; a failing call leaves its message for its own thread only."""
    from concurrent.futures import ThreadPoolExecutor

    streams = []
    for k in range(8):
        x = tone(20000 + 997 * k, 1 + k % 2, 16, seed=40 + k)
        streams.append((x, F.encode(x, 16000, subframe=("auto", "lpc", "fixed2", "verbatim")[k % 4], blocksize=1152)))

    def work(k):
        x, enc = streams[k % 8]
        for _ in range(6):
            got, sr = decode_flac(enc, verify_md5=True)
            assert sr == 16000 and np.array_equal(got, as_decoded(x, 16))
        if k % 3 == 0:
            try:
                decode_flac(b"not a flac stream at all")
            except L.VtcError as e:
                assert "fLaC" in str(e)
                return "failed as expected"
        return "ok"

    with ThreadPoolExecutor(max_workers=8) as pool:
        out = list(pool.map(work, range(24)))
    assert out.count("failed as expected") == 8 and out.count("ok") == 16


def test_threaded_decode_equals_sequential():
    """vtc_flac_decode_threads: pieces of a long stream decoded by several threads must give exactly the sequential
    result -- on noise stored verbatim (the payload is full of byte pairs that look like a frame sync), on a
    predictor-coded stereo stream, on a variable-block-size stream (frames carry sample numbers) and on a stream
    without a declared length (sequential fall-back)."""
    rng = np.random.default_rng(77)
    noise = rng.integers(-32768, 32768, size=(3_200_000, 1), dtype=np.int64)
    noise[::4097] = -8          # 0xFFF8: a sync look-alike in every frame, besides the random ones
    cases = [("verbatim noise", noise, dict(subframe="verbatim", blocksize=4096)),
             ("coded stereo", tone(1_600_000, 2, 16, seed=9), dict(subframe="fixed2", blocksize=4608, stereo="mid_side")),
             ("variable blocks", noise[:2_400_000], dict(subframe="verbatim", variable_blocks=[4096, 1152, 576, 4608])),
             ("no declared length", noise[:1_700_000], dict(subframe="verbatim", blocksize=4096, with_total=False))]
    lib = L.load()
    for name, x, kw in cases:
        enc = F.encode(x, 16000, **kw)
        assert len(enc) > 3 * (1 << 20), name
        want, sr = decode_flac(enc, verify_md5=True, n_threads=1)
        assert np.array_equal(want, as_decoded(x, 16)), name
        for nt in (2, 3, 5, 64):
            got, _ = decode_flac(enc, n_threads=nt)
            assert got.dtype == want.dtype and np.array_equal(got, want), (name, nt)
        # a damaged frame in the middle is reported, not skipped
        bad = bytearray(enc)
        bad[len(bad) // 2] ^= 0x40
        n, out = C.c_int64(), np.empty((len(x), x.shape[1]), dtype=np.int16)
        rc = lib.vtc_flac_decode_threads(bytes(bad), len(bad), C.c_void_p(out.ctypes.data), 0, len(x), C.byref(n), 4)
        assert rc == L.ERR_ARG, name


def _ogg_pages(packets, serial=0x1234, max_page=4000, header_packets=0):
    """Test muxer (RFC 3533): packets -> pages of at most max_page body bytes, packets continued across pages; the first
    `header_packets` + 1 packets get pages of their own, as the FLAC mapping asks."""
    from vtc_amd.audio import ogg_crc32
    pages, seq = [], 0
    cur_table, cur_body, cont = [], b"", False

    def flush(first_flag_cont, last=False):
        nonlocal cur_table, cur_body, seq
        if not cur_table:
            return
        flags = (1 if first_flag_cont else 0) | (2 if seq == 0 else 0) | (4 if last else 0)
        hdr = b"OggS" + struct.pack("<BBqIIIB", 0, flags, 0, serial, seq, 0, len(cur_table)) + bytes(cur_table)
        page = bytearray(hdr + cur_body)
        page[22:26] = struct.pack("<I", ogg_crc32(page))
        pages.append(bytes(page))
        seq += 1
        cur_table, cur_body = [], b""

    page_cont = False
    for k, pkt in enumerate(packets):
        segs = [pkt[i:i + 255] for i in range(0, len(pkt), 255)]
        if not segs or len(segs[-1]) == 255:
            segs.append(b"")
        for j, sg in enumerate(segs):
            if len(cur_table) == 255 or (cur_table and len(cur_body) + len(sg) > max_page):
                flush(page_cont)
                page_cont = j > 0           # the new page starts in the middle of this packet
            cur_table.append(len(sg))
            cur_body += sg
        if k <= header_packets:
            flush(page_cont)
            page_cont = False
    flush(page_cont, last=True)
    return b"".join(pages)


def _flac_frames(stream):
    """Split a native FLAC stream written by the test encoder into (metadata blocks, frames): a frame ends where the CRC-16
    of everything since its start is followed by the next sync code (or the end of the stream)."""
    assert stream[:4] == b"fLaC"
    pos, blocks = 4, []
    while True:
        last, size = stream[pos] & 0x80, int.from_bytes(stream[pos + 1:pos + 4], "big")
        blocks.append(stream[pos:pos + 4 + size])
        pos += 4 + size
        if last:
            break
    frames, start = [], pos
    while start < len(stream):
        end = start + 6
        while True:
            nxt = stream.find(b"\xff\xf8", end) if stream[start + 1] == 0xF8 else stream.find(b"\xff\xf9", end)
            cand = len(stream) if nxt < 0 else nxt
            if F.crc16(stream[start:cand - 2]) == int.from_bytes(stream[cand - 2:cand], "big"):
                frames.append(stream[start:cand])
                start = cand
                break
            assert nxt >= 0, "frame boundary not found"
            end = nxt + 1
    return blocks, frames


def test_ogg_container_crc_flac_mapping_and_codec_names(tmp_path):
    """`.ogg` is on the reference's manifest list (scripts/make_manifest.py:26-39).  The container is parsed (RFC 3533); FLAC in
    Ogg (RFC 9639 section 10.1) is decoded through the library's FLAC decoder; Vorbis / Opus streams raise with the codec's name.
    The page CRC (not zlib's CRC-32) is pinned by a real Ogg Vorbis file, a 0.5 s interface sound shipped with MathJax
    (Apache-2.0), copied to tests/golden/ogg_vorbis_beep.ogg: all of its pages must verify."""
    from vtc_amd.audio import decode_ogg, ogg_crc32, ogg_packets
    real = (GOLDEN / "ogg_vorbis_beep.ogg").read_bytes()
    pkts = ogg_packets(real, check_pages=-1)                       # raises if any page fails its CRC
    assert len(pkts) == 29 and bytes(pkts[0][:7]) == b"\x01vorbis" and bytes(pkts[1][:7]) == b"\x03vorbis"
    damaged = bytearray(real)
    damaged[100] ^= 0x10
    with pytest.raises(ValueError, match="CRC"):
        ogg_packets(bytes(damaged), check_pages=-1)
    assert ogg_crc32(b"") == 0 and ogg_crc32(b"\x01") == 0x04C11DB7   # the polynomial itself, unreflected
    (tmp_path / "beep.ogg").write_bytes(real)
    with pytest.raises(ValueError, match="carries Vorbis"):
        read_pcm(tmp_path / "beep.ogg")
    opus_head = b"OpusHead" + bytes([1, 1]) + struct.pack("<HIhB", 312, 48000, 0, 0)
    (tmp_path / "o.opus").write_bytes(_ogg_pages([opus_head, b"OpusTags" + bytes(8), bytes(40)]))
    with pytest.raises(ValueError, match="carries Opus"):
        read_pcm(tmp_path / "o.opus")

    # FLAC in Ogg: mono and stereo, small pages so that frames continue across pages, with and without the header count
    for nch, sr, page, declared in ((1, 16000, 4000, True), (2, 44100, 700, True), (1, 8000, 60000, False)):
        x = tone(30000, nch, 16, seed=nch)
        native = F.encode(x, sr, blocksize=1152 if nch == 2 else 4096)
        blocks, frames = _flac_frames(native)
        assert len(frames) == -(-30000 // (1152 if nch == 2 else 4096))
        ident = b"\x7fFLAC" + bytes([1, 0]) + (len(blocks) - 1 if declared else 0).to_bytes(2, "big") + b"fLaC" + blocks[0]
        ogg = _ogg_pages([ident] + blocks[1:] + frames, max_page=page, header_packets=len(blocks) - 1)
        got, got_sr = decode_ogg(ogg)
        assert got_sr == sr and np.array_equal(got, as_decoded(x, 16))
        p = tmp_path / f"f{nch}_{page}.oga"
        p.write_bytes(ogg)
        again, _ = read_pcm(p)
        assert np.array_equal(again, got)
        bad = bytearray(ogg)
        bad[len(ogg) // 2] ^= 0x01                                   # inside an audio page: the FLAC frame CRC catches it
        with pytest.raises(Exception):
            decode_ogg(bytes(bad))
