#!/usr/bin/env python3
"""End-to-end rate of the reference's use case: FASTA files on disk -> CSV rows
("report kernel-resident and end-to-end separately").
Writes M synthetic alignments (N x L, one line per sequence, like the reference's
examples) to a scratch directory, then times
    python -m polyfasta_amd -d DIR -o out.csv -s
in-process, and the host-buffer scan (H2D + kernel + D2H, PCIe-inclusive)."""
import argparse
import ctypes
import io
import json
import os
import shutil
import sys
import tempfile
import time
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import _lib, cli, scan, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=2000)
    ap.add_argument("--rows", type=int, default=50)
    ap.add_argument("--sites", type=int, default=10000)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    lib = _lib.load()
    _lib.init(0)
    N, L, M = args.rows, args.sites, args.files
    p = synth.make_params(N, 0.01)
    stride = scan.row_stride_for(L, 128)
    sp = _lib.SynthParamsC(seed=7, first_aln=0, n_aln=M, n_rows=N, n_sites=L, row_stride=stride,
                           seg_thresh=p.seg_thresh, tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
    bases = np.zeros(M * N * stride, dtype=np.uint8)
    _lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                       _lib.ptr(np.ascontiguousarray(p.units))))
    report = {"files": M, "rows": N, "sites": L, "host_cores": os.cpu_count()}

    # ---- PCIe-inclusive: host bases -> results through pfa_scan_batch
    descs = np.zeros(M, dtype=_lib.DESC_DTYPE)
    descs["base_off"] = np.arange(M, dtype=np.uint64) * np.uint64(N * stride)
    descs["n_rows"], descs["n_sites"], descs["row_stride"], descs["mat_rows"] = N, L, stride, N
    descs["row_idx_off"] = _lib.PFA_NO_ROW_IDX
    t = scan.finalize_layout(descs, None, len(bases))
    res = np.zeros(M, dtype=_lib.RESULT_DTYPE)
    sfs = np.zeros(M * (N // 2), dtype=np.uint32)
    for rep in range(3):
        t0 = time.perf_counter()
        _lib.check(lib.pfa_scan_batch(_lib.ptr(bases), len(bases), _lib.ptr(t.descs), M, None, 0,
                                      _lib.ptr(res), _lib.ptr(sfs), None, None, None, None))
        dt = time.perf_counter() - t0
    report["host_buffer_scan"] = {"alignments_per_s": M / dt, "GBps_incl_pcie": M * N * L / dt / 1e9, "seconds": dt}

    # ---- files -> CSV
    d = args.dir or tempfile.mkdtemp(prefix="pfa_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    os.makedirs(d, exist_ok=True)
    hdr = [f">indiv{i + 1}\n".encode() for i in range(N)]
    t0 = time.perf_counter()
    for m in range(M):
        mat = bases[m * N * stride:(m + 1) * N * stride].reshape(N, stride)[:, :L]
        with open(os.path.join(d, f"aln{m:06d}.fa"), "wb") as fh:
            fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
    report["write_files_s"] = time.perf_counter() - t0
    out = os.path.join(d, "_out.csv")
    argv = ["-f"] + [os.path.join(d, f"aln{m:06d}.fa") for m in range(M)] + ["-o", out, "-s"]
    if args.threads:
        os.environ["PFA_THREADS"] = str(args.threads)
    for rep in range(2):
        if os.path.exists(out):
            os.unlink(out)
        buf = io.StringIO()
        t0 = time.perf_counter()
        with redirect_stdout(buf):
            cli.main(argv)
        dt = time.perf_counter() - t0
    rows = open(out).read().splitlines()
    ok = len(rows) == M and all(int(r.split(",")[4]) == int(res[i]["seg_sites"]) for i, r in enumerate(rows))
    report["files_to_csv"] = {"alignments_per_s": M / dt, "seconds": dt, "rows_ok": ok,
                              "text_GBps": M * N * (L + 10) / dt / 1e9}
    print(json.dumps(report))
    if not args.dir:
        shutil.rmtree(d)


if __name__ == "__main__":
    main()
