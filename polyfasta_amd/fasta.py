"""FASTA ingest with the reference reader's record semantics
(``readfasta``, PolyFastA.py:227-250), done by the native packer of
libpfa_hip.so (csrc/pfa_fasta.cpp): text bytes -> row-major upper-case byte
matrix, one byte per base, ready for upload.
"""
from __future__ import annotations

import sys
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import FASTA_REC_DTYPE
from .scan import row_stride_for


@dataclass
class Alignment:
    """What ``readfasta`` returns as a dict, in matrix form."""
    names: list                # headers in dict order (first occurrence)
    seq_lens: np.ndarray       # uint32 per record
    matrix: np.ndarray = None  # uint8 [N, L] view (row stride >= L), None when lengths differ
    ragged: list = None        # per-record uint8 arrays when lengths differ

    @property
    def n_rows(self) -> int:
        return len(self.names)

    @property
    def same_length(self) -> bool:
        return bool((self.seq_lens == self.seq_lens[0]).all()) if len(self.seq_lens) else True

    def to_dict(self):
        """dict header -> sequence string, exactly what ``readfasta`` returns."""
        rows = self.matrix if self.matrix is not None else self.ragged
        return {n: bytes(row).decode("latin-1") for n, row in zip(self.names, rows)}


def parse_fasta_bytes(data, stride_align: int = 16):
    """-> Alignment, or None where the reference prints "is not FASTA" and
    returns 1 (no header line, or the last header line is empty)."""
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    buf = np.ascontiguousarray(buf)
    n = lib.pfa_fasta_index(_lib.ptr(buf) if len(buf) else None, len(buf), None, 0)
    if n == _lib.PFA_ERR_FORMAT and "not FASTA" in _lib.last_error():
        return None
    _lib.check(n)
    recs = np.zeros(max(n, 1), dtype=FASTA_REC_DTYPE)
    _lib.check(lib.pfa_fasta_index(_lib.ptr(buf), len(buf), _lib.ptr(recs), n))
    recs = recs[:n]
    raw = buf.tobytes() if n < 4096 else None
    names = []
    for r in recs:
        o, ln = int(r["name_off"]), int(r["name_len"])
        names.append((raw[o:o + ln] if raw is not None else buf[o:o + ln].tobytes()).decode("utf-8", "surrogateescape"))
    lens = recs["seq_len"].astype(np.uint32)
    aln = Alignment(names=names, seq_lens=lens)
    if n and aln.same_length:
        L = int(lens[0])
        stride = row_stride_for(L, stride_align)
        mat = np.empty((n, stride), dtype=np.uint8)
        _lib.check(lib.pfa_fasta_pack(_lib.ptr(buf), len(buf), _lib.ptr(np.ascontiguousarray(recs)), n,
                                      _lib.ptr(mat), stride))
        aln.matrix = mat[:, :L]
    elif n:
        aln.ragged = []
        for i in range(n):
            L = int(lens[i])
            row = np.empty(row_stride_for(L), dtype=np.uint8)
            _lib.check(lib.pfa_fasta_pack(_lib.ptr(buf), len(buf), _lib.ptr(np.ascontiguousarray(recs[i:i + 1])), 1,
                                          _lib.ptr(row), len(row)))
            aln.ragged.append(row[:L])
    return aln


def load_files(paths, bases_out: np.ndarray, stride_align: int = 128, want_names: bool = False, threads: int = 0):
    """Native multi-threaded ingest of many files into `bases_out` (uint8 buffer).
    Returns (info FILE_INFO_DTYPE[len(paths)], names or None, bytes_used).
    info.status == PFA_FILE_DEFERRED marks files that did not fit this call's
    buffer.  `names[i]` is the list of record names of file i, as bytes (want_names)."""
    import ctypes
    import os
    lib = _lib.load()
    n = len(paths)
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    info = np.zeros(n, dtype=_lib.FILE_INFO_DTYPE)
    names_buf = np.zeros(64 * 1024 * 1024 if want_names else 1, dtype=np.uint8)
    used, nused = ctypes.c_uint64(), ctypes.c_uint64()
    threads = threads or int(os.environ.get("PFA_THREADS", "0")) or min(16, os.cpu_count() or 1)
    _lib.check(lib.pfa_fasta_load_files(arr, n, stride_align, threads, _lib.ptr(bases_out), bases_out.nbytes,
                                        _lib.ptr(info), _lib.ptr(names_buf) if want_names else None, names_buf.nbytes,
                                        ctypes.byref(used), ctypes.byref(nused)))
    names = None
    if want_names:
        # header names of file i: k NUL-terminated byte strings from names_off[i]; the files' name blocks
        # sit back to back in arbitrary order, so a block ends where the next one (by offset) begins.
        # They stay bytes: --pops matches substrings on them without decoding 100 k x N headers.
        raw = names_buf[:nused.value].tobytes()
        rec = info.tolist()
        live = sorted((fi[5], i) for i, fi in enumerate(rec) if fi[1] and fi[0] != _lib.PFA_FILE_DEFERRED)
        ends = {i: (live[j + 1][0] if j + 1 < len(live) else len(raw)) for j, (_, i) in enumerate(live)}
        names = []
        for i, fi in enumerate(rec):
            if i not in ends:
                names.append([])
                continue
            names.append(raw[fi[5]:ends[i]].split(b"\0")[:fi[1]])
    return info, names, used.value


class PinnedBuffer:
    """Page-locked host buffer (hipHostMalloc) exposed as a numpy uint8 array."""

    def __init__(self, nbytes: int):
        import ctypes
        _lib.init()
        p = ctypes.c_void_p()
        _lib.check(_lib.load().pfa_host_alloc(int(nbytes), ctypes.byref(p)))
        self.ptr = p
        self.array = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(int(nbytes),))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            self.array = None
            _lib.load().pfa_host_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def read_fasta_file(path: str, stride_align: int = 16):
    with open(path, "rb") as fh:
        return parse_fasta_bytes(fh.read(), stride_align)


def read_fasta_stdin(stride_align: int = 16):
    return parse_fasta_bytes(sys.stdin.buffer.read(), stride_align)


def read_text_alignment(lines):
    """The reference's reader on text lines, call for call (PolyFastA.py:231-236 / :240-245): a character is a site whatever
    its UTF-8 length, `.upper()` is Unicode's (one sharp s becomes two letters) and `.rstrip()` strips Unicode white space.
    Returns the dict header -> sequence, or None where the reference returns 1 (no header line).  The native loader keeps
    one byte per base and hands files with non-ASCII sequence characters back; they are read with this."""
    data = {}
    head = ""
    for line in lines:
        if line[0] == ">":
            head = line[1:].rstrip()
            data[head] = ""
        elif len(head) != 0 and line[0] != ">":
            data[head] += line.rstrip().upper()
    return None if head == "" else data


def byte_codes(seqs):
    """str.translate table that gives every distinct non-ASCII character of `seqs` a byte value of its own: the scan only asks
    whether two rows hold the same symbol and whether a symbol is exactly A, C, G or T, so any value no other symbol of the
    alignment uses will do — 0x80 ... 0xFF first, then the ASCII values the alignment leaves free (never A, C, G, T, never 0:
    the padding of matrix_from_dict).  None when the alignment holds more distinct symbols than a byte has values (one byte
    per base is the layout of this implementation: well over 200 distinct characters in one alignment)."""
    chars = {c for x in seqs for c in x}
    others = sorted(c for c in chars if ord(c) >= 0x80)
    if len(others) <= 128:
        return {ord(c): 0x80 + i for i, c in enumerate(others)}
    used = {ord(c) for c in chars if ord(c) < 0x80} | set(b"ACGT") | {0}
    free = list(range(0x80, 0x100)) + [v for v in range(1, 0x80) if v not in used]
    if len(others) > len(free):
        return None
    return {ord(c): v for c, v in zip(others, free)}


def matrix_from_dict(d, seqlen=None) -> np.ndarray:
    """uint8 [N, seqlen] from the reference's in-memory alignment (dict header ->
    string).  Non-ASCII characters get byte values of their own (`byte_codes`); a sequence shorter than `seqlen`
    is padded with NUL, which plays the role of the empty slice ``''`` that the
    reference would see there (PolyFastA.py:256)."""
    seqs = list(d.values())
    if seqlen is None:
        seqlen = len(seqs[0]) if seqs else 0
    table = byte_codes(s[:seqlen] for s in seqs)
    if table is None:
        raise ValueError("more distinct characters than a byte has values: they cannot be packed one byte per base")
    out = np.zeros((len(seqs), seqlen), dtype=np.uint8)
    for i, s in enumerate(seqs):
        b = s[:seqlen].translate(table).encode("latin-1")
        out[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return out
