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
    fa, fb = got.split(","), want.split(",")
    if len(fa) != len(fb) or len(fa) != (14 if cds else 8):
        return "column count"
    names = ("file,sites_S,sites_N,pop,N,seg_sites_S,seg_sites_N,pi_S,pi_N,theta_S,theta_N,tajimasD_S,tajimasD_N,nstops" if cds
             else "file,seqlen,pop,N,seg_sites,pi,theta,tajimasD").split(",")
    for name, x, y in zip(names, fa, fb):
        if x == y:
            continue
        if not name.startswith(("pi", "theta", "tajimasD", "sites_")):
            return name
        try:
            fx, fy = float(x), float(y)
        except ValueError:
            return name
        tol = 1e-6 if (jc and name.startswith("pi")) else 1e-9
        if abs(fx - fy) <= tol * max(abs(fx), abs(fy)) or (name.startswith("tajimasD") and abs(fx - fy) <= 1e-9):
            continue
        # sites_S / sites_N are printed as round(x, 2) of a sum the reference accumulates column by column in floating point
        # (PolyFastA.py:307, :167): when the exact sum is a tie at two decimals (k/8 steps: ....375) the last bits of that
        # accumulation decide the printed digit.  One unit in the second decimal is the same value; callers count these.
        if name.startswith("sites_") and abs(fx - fy) <= 0.010000001:
            continue
        return name
    return None


def rounding_ties(got, want):
    """how many of the two `sites_*` columns differ (by one unit in the second decimal) between two coding rows"""
    fa, fb = got.split(","), want.split(",")
    return sum(1 for x, y in zip(fa[1:3], fb[1:3]) if x != y)


def sha1_ints(xs):
    import hashlib
    return hashlib.sha1(np.asarray(xs, dtype=np.int64).tobytes()).hexdigest()
