"""Host sanitizers over the decoders that parse untrusted files (csrc/flac_decode.h, csrc/adpcm_decode.h): a short mutation
campaign of tools/fuzz_decoders.cpp built with -fsanitize=address,undefined, once as shipped (frame CRCs stop most mutants at
the first damaged frame) and once with the CRC checks compiled out so that damaged frames reach the subframe parsers.
Any out-of-bounds access, signed overflow or leak ends the run with a report.  (Longer campaigns: DESIGN.md section 8.)"""

import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import flac_codec as FL

ROOT = Path(__file__).resolve().parent.parent


def _seeds(tmp_path):
    rng = np.random.default_rng(1)
    t = np.arange(16000) / 16000
    m = np.clip(np.round((0.4 * np.sin(2 * np.pi * 220 * t) + 0.02 * rng.standard_normal(len(t))) * 32767), -32768, 32767).astype(np.int64)
    streams = {
        "a": FL.encode(m[:9000], 16000),
        "b": FL.encode(m[:5000], 16000, subframe="lpc"),
        "c": FL.encode(np.stack([m[:6000], m[:6000] // 2], axis=1), 44100),
        "d": FL.encode(m[:4000] * 200, 16000, bps=24, wide_rice=True, escape=True),
        "e": FL.encode(m[:9000], 8000, blocksize=1152, with_total=False, porder=4, subframe="fixed2"),
        "f": FL.encode(m[:3000] // 256, 16000, bps=8, variable_blocks=[500, 1000, 1500]),
        "g": FL.encode(np.stack([m[:3000], m[:3000]], axis=1), 16000, stereo="mid_side", subframe="verbatim"),
    }
    paths = []
    for k, v in streams.items():
        p = tmp_path / f"{k}.flac"
        p.write_bytes(v)
        paths.append(str(p))
    return paths


@pytest.mark.parametrize("skip_crc", [False, True])
def test_decoders_survive_mutated_streams_under_asan_and_ubsan(tmp_path, skip_crc):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "fuzz"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
           "-I", str(ROOT / "vtc_amd" / "csrc"), str(ROOT / "tools" / "fuzz_decoders.cpp"), "-o", str(exe)]
    if skip_crc:
        cmd.insert(1, "-DVTC_FLAC_FUZZ_SKIP_CRC")
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "ubsan" in b.stderr.lower() or "sanitize" in b.stderr.lower()):
        pytest.skip("host sanitizer runtimes are not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe), "3000", "11"] + _seeds(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "no sanitizer report" in r.stdout
