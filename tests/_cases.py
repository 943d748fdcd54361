"""Shared helpers: rebuild fixture inputs from their generator parameters."""
import json
import os

import numpy as np

from polyfasta_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def matrix_from_gen(g):
    """uint8 [N, L] of a fixture's `gen` block (same recipe as gen_golden.gen_case)."""
    params = synth.make_params(g["N"], g["theta"])
    m = synth.synth_alignment(g["seed"], g["aln"], g["L"], params, coding=g.get("coding", False))
    inj = g.get("inject")
    if inj:
        m = synth.inject_symbols(m, inj["seed"], inj["frac"], inj["alphabet"].encode())
    return m


def g10_matrix(g):
    """uint8 [N, L] of a G10 (many rows) case: the same recipe as gen_golden.g10_matrix"""
    if g["kind"] == "many":
        m = synth.many_allele_matrix(g["seed"], g["N"], g["L"], g["alphabet"].encode())
    else:
        m = synth.synth_alignment(g["seed"], 0, g["L"], synth.make_params(g["N"], g["theta"]), coding=g.get("coding", False))
        if g.get("inject"):
            m = synth.inject_symbols(m, g["inject"]["seed"], g["inject"]["frac"], g["inject"]["alphabet"].encode())
    if g.get("subset"):
        m = np.ascontiguousarray(m[synth.row_subset(g["subset"]["seed"], g["N"], g["subset"]["frac"])])
    return m


def g10_full_matrix_and_rows(g):
    """(matrix of ALL rows, row subset or None): what a --pops unit of a huge alignment looks like to the ABI"""
    if not g.get("subset"):
        return g10_matrix(g), None
    full = dict(g)
    sub = full.pop("subset")
    return g10_matrix(full), synth.row_subset(sub["seed"], g["N"], sub["frac"])


def g12_matrix(g):
    """uint8 [N, L] of a G12 (rounding ties) case: the same recipe as gen_golden.g12_matrix"""
    m = synth.synth_alignment(g["seed"], g["aln"], g["L"], synth.make_params(g["N"], g["theta"]), coding=True)
    if g.get("inject"):
        m = synth.inject_symbols(m, g["inject"]["seed"], g["inject"]["frac"], g["inject"]["alphabet"].encode())
    for pl in g.get("plant", []):
        rows = None if pl.get("rows") is None else np.arange(pl["rows"][0], g["N"], pl["rows"][1])
        m = synth.plant_codons(m, pl["col"], pl["codons"], rows)
    return m


def codon_sets_numpy(m):
    """what pfa_plan_codon_sets returns for the matrix `m` (uint8 [N, L]), by numpy: per codon column the set of distinct
    clean codons, bit 16 e0 + 4 e1 + e2 with A=0 C=1 T=2 G=3; 0 for the partial last column (PolyFastA.py:291, :301)"""
    n, L = m.shape
    code = np.full(256, -1, dtype=np.int64)
    for ch, v in zip(b"ACTG", range(4)):
        code[ch] = v
    e = code[m[:, :L // 3 * 3]].reshape(n, -1, 3)
    clean = (e >= 0).all(axis=2)
    idx = (e[:, :, 0] * 16 + e[:, :, 1] * 4 + e[:, :, 2]).astype(np.uint64)
    bits = np.where(clean, np.uint64(1) << np.where(clean, idx, np.uint64(0)), np.uint64(0))
    s = np.bitwise_or.reduce(bits, axis=0) if n else np.zeros(L // 3, dtype=np.uint64)
    return np.concatenate([s, np.zeros(1, dtype=np.uint64)]) if L % 3 else s


def syn_bins_numpy(m):
    """uint32[65]: the bins of synonymous-site numerators of `m` (include/pfa.h: syn_num), by numpy"""
    from polyfasta_amd import _lib
    lib = _lib.load()
    t3 = np.array([lib.pfa_codon_syn_thirds(c) for c in range(64)], dtype=np.int64)
    bins = np.zeros(65, dtype=np.int64)
    for s in codon_sets_numpy(m).tolist():
        if s:
            cods = [c for c in range(64) if (s >> c) & 1]
            bins[len(cods)] += int(t3[cods].sum())
    return bins.astype(np.uint32)


def sparse_sfs(sfs):
    """a folded SFS as the G10 records hold it"""
    return {"len": len(sfs), "nz": [[i, int(c)] for i, c in enumerate(sfs) if c]}


def dict_from_gen(g):
    return synth.to_dict(synth.headers_for(g["N"], g.get("pops", 0)), matrix_from_gen(g))


def matrix_from_literal(d):
    return np.array([np.frombuffer(s.encode(), dtype=np.uint8) for s in d.values()])


def close(a, b, rel):
    if isinstance(a, str) or isinstance(b, str):
        return a == b
    return a == b or abs(a - b) <= rel * max(abs(a), abs(b))


def rows_close(got, want, rel=1e-9, rel_jc=1e-6, jc=False, abs_d=1e-9):
    """Compare two CSV rows field by field: text equal, or numerically close.
    Tajima's D (a cancelling difference) also passes on a small absolute gap."""
    fa, fb = got.split(","), want.split(",")
    assert len(fa) == len(fb), (got, want)
    for x, y in zip(fa, fb):
        if x == y:
            continue
        fx, fy = float(x), float(y)
        tol = rel_jc if jc else rel
        assert abs(fx - fy) <= tol * max(abs(fx), abs(fy)) or abs(fx - fy) <= abs_d, (got, want)
    return True


def row_matches(got, want, cds, jc):
    """One CSV row, column by column: text columns and integers equal; theta and D to 1e-9 relative (D, a cancelling
    difference, also 1e-9 absolute); pi to 1e-9 — 1e-6 when Jukes-Cantor corrected (BASELINE.json).  Returns None or the
    name of the first column that differs."""
    ncols = 14 if cds else 8
    fa, fb = got.rsplit(",", ncols - 1), want.rsplit(",", ncols - 1)      # (from the right: a file name may hold commas)
    if len(fa) != len(fb) or len(fa) != ncols:
        return "column count"
    names = ("file,sites_S,sites_N,pop,N,seg_sites_S,seg_sites_N,pi_S,pi_N,theta_S,theta_N,tajimasD_S,tajimasD_N,nstops" if cds
             else "file,seqlen,pop,N,seg_sites,pi,theta,tajimasD").split(",")
    for name, x, y in zip(names, fa, fb):
        if x == y:
            continue
        if not name.startswith(("pi", "theta", "tajimasD")):      # (sites_S / sites_N: the printed text itself, tests/golden G12)
            return name
        try:
            fx, fy = float(x), float(y)
        except ValueError:
            return name
        tol = 1e-6 if (jc and name.startswith("pi")) else 1e-9
        if abs(fx - fy) <= tol * max(abs(fx), abs(fy)) or (name.startswith("tajimasD") and abs(fx - fy) <= 1e-9):
            continue
        return name
    return None


def sha1_ints(xs):
    import hashlib
    return hashlib.sha1(np.asarray(xs, dtype=np.int64).tobytes()).hexdigest()
