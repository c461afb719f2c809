"""Manifest, sharding, resume and output helpers of the VTC step.

Behavioural mirror of the parts of the reference's src/utils.py that `src/pipeline/vtc.py` calls
(polars is not available here; tables are pyarrow, columns are numpy):
  shard_list .............. src/utils.py:271-285
  load_manifest ........... src/utils.py:293-320
  sample_manifest ......... src/utils.py:229-268
  merge_segments .......... src/utils.py:323-390
  atomic_write_parquet .... src/utils.py:393-401
  log_benchmark ........... src/utils.py:434-471
  load_completed_ids ...... src/utils.py:497-526
  get_dataset_paths ....... src/utils.py:72-175
"""

from __future__ import annotations

import argparse
import json
import os
import platform
import random
import socket
import time
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

MANIFEST_EXTENSIONS = [".parquet", ".csv", ".tsv", ".xlsx", ".xls", ".json", ".jsonl"]   # src/utils.py:40-48
BENCHMARK_LOG = Path("logs/benchmarks.jsonl")

SEGMENT_SCHEMA = pa.schema([("uid", pa.string()), ("onset", pa.float64()), ("offset", pa.float64()),
                            ("duration", pa.float64()), ("label", pa.string())])
META_SCHEMA = pa.schema([("uid", pa.string()), ("vtc_threshold", pa.float64()), ("vtc_speech_dur", pa.float64()),
                         ("vtc_n_segments", pa.int64()), ("vtc_label_counts", pa.string()),
                         ("vtc_max_sigmoid", pa.float64()), ("vtc_mean_sigmoid", pa.float64()), ("error", pa.string())])


def set_seeds(seed: int = 42) -> None:
    random.seed(seed)
    np.random.seed(seed)
    try:
        import torch
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
    except ImportError:
        pass


def hhmmss(seconds: float) -> str:
    s = int(seconds)
    return f"{s // 3600:02d}:{s % 3600 // 60:02d}:{s % 60:02d}"


# --------------------------------------------------------------------------- paths / manifest
@dataclass
class DatasetPaths:
    dataset: str
    manifest: Path
    output: Path
    figures: Path


def resolve_manifest(arg: str) -> Path:
    """An explicit manifest path, or a bare dataset name looked up as manifests/{name}.<ext>
    (.csv wins when several formats exist)."""
    p = Path(arg)
    if "/" in arg or "\\" in arg or p.suffix.lower() in MANIFEST_EXTENSIONS:
        if not p.exists():
            raise FileNotFoundError(f"Manifest file not found: {p.resolve()}")
        if p.suffix.lower() not in MANIFEST_EXTENSIONS:
            raise ValueError(f"Unsupported manifest format '{p.suffix}'. Supported: {', '.join(MANIFEST_EXTENSIONS)}")
        return p
    root = Path("manifests")
    if not root.is_dir():
        raise FileNotFoundError(f"No 'manifests/' directory found and '{arg}' is not a path to an existing file.")
    hits = [root / f"{arg}{ext}" for ext in MANIFEST_EXTENSIONS if (root / f"{arg}{ext}").is_file()]
    if not hits:
        raise FileNotFoundError(f"No manifest found for dataset '{arg}' (searched manifests/{arg}.<ext>)")
    if len(hits) == 1:
        return hits[0]
    csv = [h for h in hits if h.suffix.lower() == ".csv"]
    if len(csv) == 1:
        return csv[0]
    raise ValueError(f"Ambiguous: multiple manifests match '{arg}': {', '.join(map(str, hits))}")


def get_dataset_paths(dataset: str) -> DatasetPaths:
    return DatasetPaths(dataset=dataset, manifest=resolve_manifest(dataset), output=Path("output") / dataset,
                        figures=Path("figures") / dataset)


def load_manifest(path: Path) -> pa.Table:
    suffix = Path(path).suffix.lower()
    if suffix == ".parquet":
        return pq.read_table(path)
    if suffix in (".csv", ".tsv"):
        from pyarrow import csv as pacsv
        return pacsv.read_csv(path, parse_options=pacsv.ParseOptions(delimiter="\t" if suffix == ".tsv" else ","))
    if suffix == ".jsonl":
        from pyarrow import json as pajson
        return pajson.read_json(path)
    if suffix == ".json":
        with open(path) as f:
            return pa.Table.from_pylist(json.load(f))
    if suffix == ".xlsx":
        return _read_xlsx(path)
    if suffix == ".xls":
        return _read_xls(path)
    raise ValueError(f"Unsupported manifest format '{suffix}'. Supported: {', '.join(MANIFEST_EXTENSIONS)}")


def _read_xlsx(path) -> pa.Table:
    """Any failure to read the workbook is a RuntimeError naming the file, as the reference's (src/utils.py:305-312)."""
    try:
        return _read_xlsx_unchecked(path)
    except RuntimeError:
        raise
    except Exception as e:      # malformed parts: missing relationship targets, bad shared-string indices, odd cell refs
        raise RuntimeError(f"Failed to read Excel file '{path}': {type(e).__name__}: {e}") from e


def _read_xlsx_unchecked(path) -> pa.Table:
    """First worksheet of an .xlsx workbook as a table (reference: pl.read_excel, src/utils.py:305-312; openpyxl is not
    needed: the container is a zip of XML parts).  First row = header; shared, inline and formula-cached strings, numbers and
    booleans; empty cells become nulls.  A column whose cells are all numbers stays numeric, anything else is text."""
    import re
    import zipfile
    import xml.etree.ElementTree as ET
    ns = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main",
          "r": "http://schemas.openxmlformats.org/officeDocument/2006/relationships",
          "p": "http://schemas.openxmlformats.org/package/2006/relationships"}
    try:
        with zipfile.ZipFile(path) as z:
            names = set(z.namelist())
            shared = []
            if "xl/sharedStrings.xml" in names:
                for si in ET.fromstring(z.read("xl/sharedStrings.xml")).findall("m:si", ns):
                    shared.append("".join(t.text or "" for t in si.iter("{%s}t" % ns["m"])))
            sheet = "xl/worksheets/sheet1.xml"
            if "xl/workbook.xml" in names and "xl/_rels/workbook.xml.rels" in names:     # the FIRST sheet in workbook order
                wb = ET.fromstring(z.read("xl/workbook.xml"))
                first = wb.find("m:sheets/m:sheet", ns)
                rid = first.get("{%s}id" % ns["r"]) if first is not None else None
                for rel in ET.fromstring(z.read("xl/_rels/workbook.xml.rels")).findall("p:Relationship", ns):
                    if rel.get("Id") == rid:
                        target = rel.get("Target").lstrip("/")
                        sheet = target if target.startswith("xl/") else "xl/" + target
            root = ET.fromstring(z.read(sheet))
    except (zipfile.BadZipFile, KeyError, ET.ParseError) as e:
        raise RuntimeError(f"Failed to read Excel file '{path}': {e}") from e

    def col_index(ref: str) -> int:
        n = 0
        for ch in re.match(r"[A-Z]+", ref).group(0):
            n = n * 26 + ord(ch) - 64
        return n - 1

    rows = []
    for row in root.findall("m:sheetData/m:row", ns):
        cells = {}
        for k, c in enumerate(row.findall("m:c", ns)):
            j = col_index(c.get("r")) if c.get("r") else k
            t, v = c.get("t", "n"), c.find("m:v", ns)
            if t == "inlineStr":
                val = "".join(x.text or "" for x in c.iter("{%s}t" % ns["m"]))
            elif v is None or v.text is None:
                val = None
            elif t == "s":
                val = shared[int(v.text)]
            elif t in ("str", "e"):
                val = v.text
            elif t == "b":
                val = v.text == "1"
            else:
                f = float(v.text)
                val = int(f) if f.is_integer() and "." not in v.text and "e" not in v.text.lower() else f
            cells[j] = val
        rows.append(cells)
    return _rows_to_table(rows)


def _rows_to_table(rows: list) -> pa.Table:
    """rows: one {column index: value} dict per sheet row, first row = header (shared by the .xlsx and .xls readers)."""
    if not rows or not any(rows):          # no rows, or rows without a single cell
        return pa.table({})
    width = max(max(r) for r in rows if r) + 1
    header = [str(rows[0].get(j)) if rows[0].get(j) is not None else f"column_{j + 1}" for j in range(width)]
    seen = {}
    for j, name in enumerate(header):      # duplicate header names get a suffix (polars: "name_duplicated_N") instead of
        if name in seen:                   # silently collapsing into one column
            header[j] = f"{name}_duplicated_{seen[name]}"
        seen[name] = seen.get(name, -1) + 1
    cols = {}
    for j, name in enumerate(header):
        vals = [r.get(j) for r in rows[1:]]
        if any(isinstance(v, str) for v in vals):
            vals = [None if v is None else (v if isinstance(v, str) else str(v)) for v in vals]
        cols[name] = vals
    return pa.table(cols)


def _read_xls(path) -> pa.Table:
    """First worksheet of a legacy .xls workbook (reference: pl.read_excel -> xlrd, src/utils.py:305-312).  Any failure is a
    RuntimeError naming the file."""
    try:
        return _rows_to_table(_xls_rows(Path(path).read_bytes()))
    except Exception as e:
        raise RuntimeError(f"Failed to read Excel file '{path}': {type(e).__name__}: {e}") from e


def _cfb_stream(blob: bytes, names) -> bytes:
    """One stream of a Compound File Binary container ([MS-CFB]: 512-byte header, FAT / DIFAT sector chains, 128-byte
    directory entries, streams under the cut-off size in the mini stream of the root entry)."""
    import struct
    if blob[:8] != bytes.fromhex("D0CF11E0A1B11AE1"):
        raise ValueError("not a compound file (bad signature)")
    sec_shift, mini_shift = struct.unpack_from("<HH", blob, 0x1E)
    n_fat, dir_start, _, cutoff, minifat_start, n_minifat, difat_start, n_difat = struct.unpack_from("<IIIIIIII", blob, 0x2C)
    ssz, msz = 1 << sec_shift, 1 << mini_shift
    END = 0xFFFFFFFE

    def sector(i):
        return blob[(i + 1) * ssz:(i + 2) * ssz]

    difat = list(struct.unpack_from("<109I", blob, 0x4C))
    nxt, hops = difat_start, 0
    while nxt not in (END, 0xFFFFFFFF) and hops < n_difat + 1:
        vals = struct.unpack(f"<{ssz // 4}I", sector(nxt))
        difat += vals[:-1]
        nxt, hops = vals[-1], hops + 1
    fat = []
    for sid in difat[:n_fat]:
        if sid < 0xFFFFFFFC:
            fat += struct.unpack(f"<{ssz // 4}I", sector(sid))

    def chain(start, table, limit):
        out, cur = [], start
        while cur < 0xFFFFFFFC:
            if cur >= len(table) or len(out) > limit:
                raise ValueError("broken sector chain")
            out.append(cur)
            cur = table[cur]
        return out

    directory = b"".join(sector(i) for i in chain(dir_start, fat, len(fat)))
    entries = []
    for off in range(0, len(directory) - 127, 128):
        nlen, typ = struct.unpack_from("<HB", directory, off + 0x40)
        if typ in (2, 5) and 2 <= nlen <= 64:
            name = directory[off:off + nlen - 2].decode("utf-16-le", "replace")
            start, size = struct.unpack_from("<IQ", directory, off + 0x74)
            entries.append((name, typ, start, size & 0xFFFFFFFF if sec_shift == 9 else size))
    root = next((e for e in entries if e[1] == 5), None)
    for want in names:
        for name, typ, start, size in entries:
            if typ == 2 and name.lower() == want.lower():
                if size >= cutoff or root is None:
                    data = b"".join(sector(i) for i in chain(start, fat, len(fat)))
                else:      # mini stream: 64-byte sectors inside the root entry's stream, chained by the mini FAT
                    mini = b"".join(sector(i) for i in chain(root[2], fat, len(fat)))
                    minifat = []
                    for sid in chain(minifat_start, fat, len(fat)) if n_minifat else []:
                        minifat += struct.unpack(f"<{ssz // 4}I", sector(sid))
                    data = b"".join(mini[i * msz:(i + 1) * msz] for i in chain(start, minifat, len(minifat)))
                return data[:size]
    raise ValueError(f"no {' / '.join(names)} stream in the file")


def _xls_rows(blob: bytes) -> list:
    """Cell values of the first worksheet of a BIFF8 workbook stream ([MS-XLS]): SST (shared strings, with CONTINUE records
    that may switch the character width mid-string), LABELSST, LABEL, NUMBER, RK, MULRK, BOOLERR and FORMULA (+ STRING)
    records; one {column: value} dict per row.  Text cells come back as str, numbers as int where they are whole."""
    import struct
    wb = _cfb_stream(blob, ("Workbook", "Book"))

    def records(pos):
        while pos + 4 <= len(wb):
            typ, ln = struct.unpack_from("<HH", wb, pos)
            yield typ, wb[pos + 4:pos + 4 + ln], pos
            pos += 4 + ln

    def num(x: float):
        return int(x) if float(x).is_integer() and abs(x) < 2 ** 53 else float(x)

    def rk(v: int):
        if v & 2:
            x = float((v >> 2) - (1 << 30 if v & 0x80000000 else 0))
        else:
            x = struct.unpack("<d", struct.pack("<Q", (v & 0xFFFFFFFC) << 32))[0]
        return num(x / 100.0 if v & 1 else x)

    def ustring(chunks, ci, off, cch_bytes=2):
        """A unicode string that starts at chunks[ci][off:]; the characters (and only they) may run on into the following
        CONTINUE chunks, each of which then opens with a fresh option byte.  Returns (text, chunk index, offset)."""
        buf = chunks[ci]
        cch = struct.unpack_from("<H" if cch_bytes == 2 else "<B", buf, off)[0]
        off += cch_bytes
        flags = buf[off]
        off += 1
        n_runs = n_ext = 0
        if flags & 8:
            n_runs = struct.unpack_from("<H", buf, off)[0]
            off += 2
        if flags & 4:
            n_ext = struct.unpack_from("<I", buf, off)[0]
            off += 4
        wide, text = flags & 1, []
        while cch:
            if off >= len(buf):
                ci += 1
                buf, off = chunks[ci], 1
                wide = buf[0] & 1
            take = min(cch, (len(buf) - off) // (2 if wide else 1))
            raw = buf[off:off + take * (2 if wide else 1)]
            text.append(raw.decode("utf-16-le", "replace") if wide else raw.decode("latin-1"))
            off += len(raw)
            cch -= take
        skip = 4 * n_runs + n_ext
        while skip:
            if off >= len(buf):
                ci += 1
                buf, off = chunks[ci], 0
            step = min(skip, len(buf) - off)
            off += step
            skip -= step
        return "".join(text), ci, off

    sst, sheets, first = [], [], None
    recs = list(records(0))
    if not recs or recs[0][0] != 0x0809 or struct.unpack_from("<H", recs[0][1], 0)[0] != 0x0600:
        raise ValueError("not a BIFF8 workbook (Excel 97-2003)")
    i = 0
    while i < len(recs):
        typ, data, _ = recs[i]
        if typ == 0x0085:                                   # BOUNDSHEET: stream offset, visibility, type, name
            sheets.append((struct.unpack_from("<I", data, 0)[0], data[5]))
        elif typ == 0x00FC:                                 # SST + its CONTINUE records
            chunks = [data]
            while i + 1 < len(recs) and recs[i + 1][0] == 0x003C:
                i += 1
                chunks.append(recs[i][1])
            n_unique = struct.unpack_from("<I", data, 4)[0]
            ci, off = 0, 8
            for _ in range(n_unique):
                if off >= len(chunks[ci]):
                    ci, off = ci + 1, 0
                s_, ci, off = ustring(chunks, ci, off)
                sst.append(s_)
        elif typ == 0x000A:                                 # EOF of the workbook globals
            break
        i += 1
    worksheets = [off for off, kind in sheets if kind == 0]
    if not worksheets:
        raise ValueError("no worksheet in the workbook")
    cells, pending = {}, None
    for typ, data, _ in records(worksheets[0]):
        if typ == 0x000A:
            break
        if typ in (0x00FD, 0x0203, 0x027E, 0x0204, 0x0205, 0x0006):
            r, c = struct.unpack_from("<HH", data, 0)
        if typ == 0x00FD:
            cells[(r, c)] = sst[struct.unpack_from("<I", data, 6)[0]]
        elif typ == 0x0203:
            cells[(r, c)] = num(struct.unpack_from("<d", data, 6)[0])
        elif typ == 0x027E:
            cells[(r, c)] = rk(struct.unpack_from("<I", data, 6)[0])
        elif typ == 0x00BD:                                 # MULRK: row, first column, (xf, rk)*, last column
            r, c0 = struct.unpack_from("<HH", data, 0)
            for k in range((len(data) - 6) // 6):
                cells[(r, c0 + k)] = rk(struct.unpack_from("<I", data, 4 + 6 * k + 2)[0])
        elif typ == 0x0204:
            cells[(r, c)] = ustring([data], 0, 6)[0]
        elif typ == 0x0205:
            if data[7] == 0:
                cells[(r, c)] = bool(data[6])
        elif typ == 0x0006:                                 # FORMULA: cached result; a string result follows in STRING
            res = data[6:14]
            if res[6:8] == b"\xff\xff":
                if res[0] == 0:
                    pending = (r, c)
                elif res[0] == 1:
                    cells[(r, c)] = bool(res[2])
            else:
                cells[(r, c)] = num(struct.unpack("<d", res)[0])
        elif typ == 0x0207 and pending is not None:
            cells[pending], pending = ustring([data], 0, 0)[0], None
    if not cells:
        return []
    rows = [dict() for _ in range(max(r for r, _ in cells) + 1)]
    for (r, c), v in cells.items():
        rows[r][c] = v
    while rows and not rows[0]:                             # leading empty rows: the header is the first row with cells
        rows.pop(0)
    return rows


def manifest_paths(table: pa.Table, column: str = "path") -> list[str]:
    if column not in table.column_names:
        raise KeyError(f"manifest has no '{column}' column (columns: {table.column_names})")
    return [p for p in table.column(column).to_pylist() if p is not None]


def sample_manifest(table: pa.Table, sample, seed: int = 42) -> pa.Table:
    """None -> unchanged; int >= 1 -> that many rows; float in (0,1) -> that fraction.  Deterministic for a
    seed (the reference uses polars' sampler, so the chosen subset differs from it; the contract is
    'fixed seed -> fixed subset')."""
    if sample is None:
        return table
    n = table.num_rows
    if isinstance(sample, float) and 0 < sample < 1:
        keep = max(1, int(round(n * sample)))
    elif isinstance(sample, (int, float)) and sample >= 1:
        keep = int(sample)
    else:
        raise ValueError(f"Invalid --sample value: {sample!r}.")
    if keep >= n:
        return table
    idx = np.sort(np.random.default_rng(seed).permutation(n)[:keep])
    return table.take(pa.array(idx))


def shard_list(items: list, array_id: int, array_count: int) -> list:
    """Contiguous slice `array_id` of `array_count`; the first len % count shards get one extra item."""
    if array_count < 1 or not (0 <= array_id < array_count):
        raise ValueError(f"array_id {array_id} not in [0, {array_count})")
    base, extra = divmod(len(items), array_count)
    lo = array_id * base + min(array_id, extra)
    return items[lo: lo + base + (1 if array_id < extra else 0)]


def balanced_shards(items: list, costs: list, n_shards: int) -> list[list]:
    """Longest-processing-time assignment by cost (e.g. file size) for skewed manifests; within a shard the
    manifest order is kept.  Not in the reference (it slices by count); used by vtc_amd.launcher."""
    order = sorted(range(len(items)), key=lambda i: (-costs[i], i))
    load = [0.0] * n_shards
    owner = [0] * len(items)
    for i in order:
        s = min(range(n_shards), key=lambda k: (load[k], k))
        owner[i] = s
        load[s] += costs[i]
    return [[items[i] for i in range(len(items)) if owner[i] == s] for s in range(n_shards)]


# --------------------------------------------------------------------------- segments
def segments_table(rows: list[dict]) -> pa.Table:
    cols = {name: [r[name] for r in rows] for name in SEGMENT_SCHEMA.names}
    return pa.table(cols, schema=SEGMENT_SCHEMA)


def meta_table(rows: list[dict]) -> pa.Table:
    cols = {name: [r[name] for r in rows] for name in META_SCHEMA.names}
    return pa.table(cols, schema=META_SCHEMA)


def merge_segments(table: pa.Table, min_duration_off_s: float = 0.1, min_duration_on_s: float = 0.1) -> pa.Table:
    """Collar merge + minimum duration, per (uid, label).

    Sorted by onset, a segment opens a new group when its onset exceeds the running maximum of all earlier
    offsets in the partition by more than the collar; a group becomes (min onset, max offset), both rounded
    to 3 decimals, duration = round(offset - onset, 3); groups shorter than min_duration_on_s are dropped.
    Result ordered by (uid, label, onset)."""
    if table.num_rows == 0:
        return table
    has_label = "label" in table.column_names
    onset = table.column("onset").to_numpy(zero_copy_only=False).astype(np.float64)
    offset = table.column("offset").to_numpy(zero_copy_only=False).astype(np.float64)

    uid_names, uid_code = _sorted_codes(table.column("uid"))
    if has_label:
        lab_names, lab_code = _sorted_codes(table.column("label"))
        part = uid_code * len(lab_names) + lab_code
    else:
        part = uid_code
    order = np.lexsort((onset, part))           # stable: ties keep input order, like a stable sort
    part, onset, offset = part[order], onset[order], offset[order]

    n = len(part)
    part_start = np.ones(n, dtype=bool)
    part_start[1:] = part[1:] != part[:-1]
    starts = np.flatnonzero(part_start)
    ends = np.append(starts[1:], n)
    prev_max = np.zeros(n, dtype=np.float64)     # first row of a partition compares against 0.0
    for a, b in zip(starts, ends):
        if b - a > 1:
            prev_max[a + 1: b] = np.maximum.accumulate(offset[a: b - 1])
    group_start = part_start | (onset > prev_max + min_duration_off_s)
    gs = np.flatnonzero(group_start)
    g_on = np.round(np.minimum.reduceat(onset, gs), 3)
    g_off = np.round(np.maximum.reduceat(offset, gs), 3)
    g_dur = np.round(g_off - g_on, 3)
    g_part = part[gs]
    keep = g_dur >= min_duration_on_s
    g_on, g_off, g_dur, g_part = g_on[keep], g_off[keep], g_dur[keep], g_part[keep]
    final = np.lexsort((g_on, g_part))
    g_on, g_off, g_dur, g_part = g_on[final], g_off[final], g_dur[final], g_part[final]
    cols = {"uid": _string_column(g_part // len(lab_names) if has_label else g_part, uid_names)}
    cols["onset"], cols["offset"], cols["duration"] = pa.array(g_on), pa.array(g_off), pa.array(g_dur)
    if has_label:
        cols["label"] = _string_column(g_part % len(lab_names), lab_names)
    return pa.table(cols)


def _sorted_codes(col):
    """String column -> (names sorted as np.unique sorts them, int64 code of every row in that order), through
    arrow's dictionary encoding instead of a string sort over all rows."""
    d = (col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col).dictionary_encode()
    names = np.asarray(d.dictionary.to_pylist(), dtype=object).astype(str)
    order = np.argsort(names, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    return names[order], rank[d.indices.to_numpy(zero_copy_only=False).astype(np.int64)]


def _string_column(codes, names) -> pa.Array:
    """names[codes] as a plain string column (built as a dictionary array, then decoded by arrow)."""
    return pa.DictionaryArray.from_arrays(pa.array(np.asarray(codes, dtype=np.int32)),
                                          pa.array([str(x) for x in names], pa.string())).cast(pa.string())


def segments_table_from_blocks(blocks, labels) -> pa.Table:
    """Per-file column blocks (uid, onset, offset, duration, label_codes) -> the segment table of segments_table()."""
    if not blocks:
        return segments_table([])
    lens = np.array([len(b[1]) for b in blocks], dtype=np.int64)
    cat = lambda k, dt: np.concatenate([np.asarray(b[k], dtype=dt) for b in blocks]) if lens.sum() else np.empty(0, dt)
    return pa.table({"uid": _string_column(np.repeat(np.arange(len(blocks)), lens), [b[0] for b in blocks]),
                     "onset": pa.array(cat(1, np.float64)), "offset": pa.array(cat(2, np.float64)),
                     "duration": pa.array(cat(3, np.float64)), "label": _string_column(cat(4, np.int32), labels)},
                    schema=SEGMENT_SCHEMA)


# --------------------------------------------------------------------------- outputs / resume
def atomic_write_parquet(table: pa.Table, path, compression: str = "zstd") -> None:
    """Write to <name>.parquet.tmp then rename, so readers never see a partial shard."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_suffix(".parquet.tmp")
    pq.write_table(table, tmp, compression=compression)
    os.replace(tmp, path)


def load_completed_ids(meta_dir: Path, id_column: str = "uid", pattern: str = "*.parquet") -> set:
    """uids present in any metadata parquet under meta_dir (error rows included: errored files are not
    retried, as in the reference)."""
    meta_dir = Path(meta_dir)
    if meta_dir.is_file():
        return set(pq.read_table(meta_dir, columns=[id_column]).column(id_column).to_pylist())
    if not meta_dir.is_dir():
        return set()
    done: set = set()
    for f in sorted(meta_dir.glob(pattern)):
        done.update(pq.read_table(f, columns=[id_column]).column(id_column).to_pylist())
    return done


def _gpu_name() -> str:
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.get_device_name(0)
    except Exception:
        pass
    return ""


def log_benchmark(step: str, dataset: str, n_files: int, wall_seconds: float, total_audio_seconds: float = 0.0,
                  total_bytes: int = 0, n_workers: int | None = None, extra: dict | None = None) -> dict:
    """Append one JSON record to logs/benchmarks.jsonl (same keys as the reference's records)."""
    rec = {
        "timestamp": time.strftime("%Y-%m-%dT%H:%M:%S%z"), "step": step, "dataset": dataset, "n_files": n_files,
        "wall_seconds": round(wall_seconds, 2), "total_audio_seconds": round(total_audio_seconds, 2),
        "total_bytes": total_bytes,
        "files_per_second": round(n_files / wall_seconds, 2) if wall_seconds > 0 else 0,
        "bytes_per_second": round(total_bytes / wall_seconds, 2) if wall_seconds > 0 and total_bytes > 0 else 0,
        "hardware": {"hostname": socket.gethostname(), "cpu_count": os.cpu_count() or 0, "n_workers": n_workers,
                     "gpu": _gpu_name(), "platform": platform.platform()},
    }
    if extra:
        rec["extra"] = extra
    BENCHMARK_LOG.parent.mkdir(parents=True, exist_ok=True)
    with open(BENCHMARK_LOG, "a") as f:
        f.write(json.dumps(rec) + "\n")
    return rec


# --------------------------------------------------------------------------- argparse
def _parse_sample(value: str):
    try:
        if "." in value:
            f = float(value)
            if 0 < f < 1:
                return f
            raise ValueError
        n = int(value)
        if n < 1:
            raise ValueError
        return n
    except ValueError:
        raise argparse.ArgumentTypeError(
            f"Invalid sample value '{value}'. Use an integer >= 1 (files) or a float in (0, 1) (fraction).")


def add_sample_argument(parser) -> None:
    parser.add_argument("--sample", default=None, type=_parse_sample, metavar="N_OR_FRAC",
                        help="Process only a random subset: integer >= 1 = that many files, float in (0,1) = "
                             "that fraction. Deterministic (seed 42).")
