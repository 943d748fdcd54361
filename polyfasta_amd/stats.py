"""Host-side finishing math: device integers -> theta_pi, theta_w, Tajima's D.

The HIP kernels return exact integers per scan unit (S, H = sum over variable
sites of N^2 - sum_j c_j^2, folded SFS, coding-sequence counters); everything
floating point is done here in double precision so that rounding behaves like
the reference's (same expressions where the reference's form matters):

    theta_pi total = N/(N-1) * sum_sites (1 - sum_j (c_j/N)^2) = H / (N (N-1))   PolyFastA.py:485-492
    theta_w total  = S / a1,  a1 = sum_{i<N} 1/i                                 PolyFastA.py:494-497
    D              = (theta_pi - theta_w) / sqrt(e1 S + e2 S (S-1))             PolyFastA.py:509,522-534
    Jukes-Cantor   = -3/4 ln(1 - 4/3 x), falling back to x when it fails         PolyFastA.py:499-500,512-516
"""
from __future__ import annotations

import math
from functools import lru_cache

NA = "NA"


@lru_cache(maxsize=None)
def harmonic_terms(n: int):
    """(a1, a2) for sample size n, summed in ascending i like the reference."""
    a1 = 0.0
    a2 = 0.0
    for i in range(1, n):
        a1 += 1.0 / i
        a2 += 1.0 / (i ** 2)
    return a1, a2


def wattersons_theta_total(seg_sites: int, n: int) -> float:
    """PolyFastA.py:494-497; ZeroDivisionError for n < 2 like the reference."""
    a1, _ = harmonic_terms(n)
    return seg_sites / a1


def pi_total(h: int, n: int) -> float:
    """Total nucleotide diversity from the integer pairwise-difference sum H."""
    return h / (n * (n - 1))


@lru_cache(maxsize=None)
def tajima_e(n: int):
    """(e1, e2) of PolyFastA.py:522-533 with the reference's operand order (they depend on n only)."""
    a1, a2 = harmonic_terms(n)
    b1 = (n + 1.0) / (3.0 * (n - 1.0))
    b2 = (2.0 * ((n ** 2.0) + n + 3.0)) / (9.0 * n * (n - 1.0))
    c1 = b1 - (1 / a1)
    c2 = b2 - ((n + 2) / (a1 * n)) + (a2 / (a1 ** 2))
    e1 = c1 / a1
    e2 = c2 / ((a1 ** 2) + a2)
    return e1, e2


def tajima_sd(seg_sites: int, n: int) -> float:
    """PolyFastA.py:534."""
    e1, e2 = tajima_e(n)
    ss = seg_sites
    return math.sqrt((e1 * ss) + (e2 * ss * (ss - 1)))


def jukes_cantor_correction(x: float) -> float:
    """PolyFastA.py:499-500."""
    return -0.75 * math.log(1 - (4.0 / 3.0) * x)


def finish_polymorphism(seg_sites: int, h: int, n: int, seqlen, jc: bool):
    """(S, pi/site, theta_w/site, D) exactly as ``polymorphism`` returns them
    (PolyFastA.py:502-520): integers 0 and the string "NA" when there is no
    variable site; D = "NA" when the variance term is zero (N = 2, 3)."""
    if seg_sites == 0:
        return 0, 0, 0, NA
    th_pi = pi_total(h, n)
    th_wa = wattersons_theta_total(seg_sites, n)
    try:
        d = (th_pi - th_wa) / tajima_sd(seg_sites, n)
    except ZeroDivisionError:
        d = NA
    if jc:
        try:
            pi_site = jukes_cantor_correction(th_pi / seqlen)
        except Exception:  # the reference uses a bare except here (PolyFastA.py:515)
            pi_site = th_pi / seqlen
    else:
        pi_site = th_pi / seqlen
    return seg_sites, pi_site, th_wa / seqlen, d


def syn_sites_from_bins(syn_num) -> float:
    """Synonymous-site count of PolyFastA.py:307 from the device's integer bins:
    sum over k of (sum of synonymous-neighbour counts of columns with k distinct
    clean codons) / (3 k)."""
    import numpy as np
    bins = np.asarray(syn_num)
    total = 0.0
    for k in np.flatnonzero(bins).tolist():      # ascending k: the summation order is part of the result
        if k:
            total += int(bins[k]) / (3.0 * k)
    return total


# ------------------------------------------------------------------ the reference's own accumulation of count_syn
# `sites_S` / `sites_N` are printed as round(x, 2) of a sum the reference builds column by column in floating point
# (PolyFastA.py:307: count_syn += sum([syncodfreq(c) for c in cod]) / len(cod); :167, :178).  The exact value of that sum —
# what the bins give — is a two-decimal tie (....375 = 3/8) whenever a codon column with 8, 16, ... distinct codons puts a
# factor 8 into its denominator; then the last bits of the reference's accumulation decide the printed digit, and they are
# reproduced here from the column's codon sets (pfa_plan_codon_sets: a re-scan of that one unit on the card).
_ALPHA_RANK = (0, 1, 3, 2)      # codon-set bit order is A C T G ((ascii >> 1) & 3); strings sort A C G T


def _syn_float_by_codon():
    """syncodfreq (PolyFastA.py:536-557) as the reference's floats, by codon-set bit: 0, 1/3, 2/3, 1., 4/3"""
    import numpy as np
    from . import _lib
    lib = _lib.load()
    f = (0.0, 1 / 3, 2 / 3, 1.0, 4 / 3)
    return np.array([f[lib.pfa_codon_syn_thirds(c)] for c in range(64)], dtype=np.float64)


_SYN_FLOAT = None


def syn_tie_candidates(syn_num, n_sites, missing_cols=None):
    """Units whose synonymous-site count (or its complement, :167) sits on a two-decimal rounding boundary as far as double
    precision can tell — or whose complement, the non-synonymous site count (seqlen - missing) - count_syn (:167), is ZERO in
    exact arithmetic (`missing_cols`: codon columns without a clean codon, per unit): the reference then divides by the
    rounding noise of its own accumulation (pi_N = 1.5e16, a negative zero, or ZeroDivisionError when the noise is nil), so
    those few units want the same last bits too (found by tools/g14_explore.py: `CTG>` / `AGGA` / `CCGT`, one clean column
    worth 1.0 of 1 site).  bool[n_units].  `syn_num`: uint32[n_units, 65]; `n_sites`: alignment length per unit.  The
    reference's accumulated rounding error is below (columns) x ulp(sum); anything within a generous multiple of that of
    xxx.xx5 is replayed.  Units whose columns all hold at most 7 distinct codons are at least 1 / (200 x 1260) = 4e-6 away
    from any boundary (the sum is a multiple of 1/1260) and are not even looked at in floating point — unless they are so
    long that the reference's own error could reach that far (millions of columns)."""
    import numpy as np
    syn = np.asarray(syn_num).reshape(-1, 65)
    n_sites = np.asarray(n_sites, dtype=np.float64)
    cols = n_sites / 3.0 + 1.0
    look = syn[:, 8:].any(axis=1) | (cols * np.spacing(np.maximum(n_sites, 1.0)) > 1e-8)
    out = np.zeros(len(syn), dtype=bool)
    if missing_cols is not None and len(syn):
        # a clean codon column is worth at most 4/3 synonymous sites of its 3: the counts can only meet when ONE clean column
        # stands against one or two counted sites (a partial last codon: seqlen - 3 x missing <= 2) — nothing to look at in an
        # ordinary batch
        counted = n_sites - 3.0 * np.asarray(missing_cols, dtype=np.float64)
        few = np.flatnonzero(counted <= 2.0)
        if len(few):
            x_few = (syn[few, 1:].astype(np.float64) / (3.0 * np.arange(1, 65, dtype=np.float64))).sum(axis=1)
            out[few] = (np.abs(counted[few] - x_few) < 1e-6) & (x_few > 0)
    if not look.any():
        return out
    idx = np.flatnonzero(look)
    bins = syn[idx].astype(np.float64)
    k3 = 3.0 * np.arange(1, 65, dtype=np.float64)
    x = (bins[:, 1:] / k3).sum(axis=1)
    y = 100.0 * x
    d = np.abs(y - np.floor(y) - 0.5)
    out[idx] |= d < 1e-6 + 400.0 * cols[idx] * np.spacing(np.maximum(x, 1.0))
    return out


def _replay_terms(s):
    """the per-column terms sum([syncodfreq(c) ...]) / len(...) of the non-empty codon sets `s` (uint64[...]); columns with three
    or more codons summed in ascending codon order"""
    import numpy as np
    global _SYN_FLOAT
    if _SYN_FLOAT is None:
        _SYN_FLOAT = _syn_float_by_codon()
    F = _SYN_FLOAT
    k = np.bitwise_count(s).astype(np.int64)
    one = np.uint64(1)
    low = s & (~s + one)                                   # lowest set bit
    c1 = np.bitwise_count(low - one).astype(np.int64)      # its index
    rest = s ^ low
    low2 = rest & (~rest + one)
    c2 = np.bitwise_count(np.where(rest != 0, low2 - one, np.uint64(0))).astype(np.int64)
    terms = np.where(k == 1, F[c1], (F[c1] + F[c2]) / 2.0)  # two terms: a + b is the same either way round
    for i in np.flatnonzero(k >= 3).tolist():
        m = int(s[i])
        cods = sorted((c for c in range(64) if (m >> c) & 1),
                      key=lambda c: (_ALPHA_RANK[c >> 4], _ALPHA_RANK[(c >> 2) & 3], _ALPHA_RANK[c & 3]))
        t = 0
        for c in cods:
            t = t + float(F[c])
        terms[i] = t / len(cods)
    return terms


def syn_sites_replay(sets):
    """count_syn exactly as getvarCDSsites accumulates it (PolyFastA.py:307), from the set of distinct clean codons of every
    codon column in column order (`sets`: uint64[ceil(L / 3)], 0 = no clean codon: the column adds nothing, :301-305).
    Per column sum([syncodfreq(c) ...]) / len(...) — Python's left-to-right float sum starting from int 0 — then added
    to the running total.  Columns with three or more codons are summed in ascending codon order (the reference walks
    a Python set of strings there: its order, and with it the last bit of such a term, follows the interpreter's hash
    seed; syn_sites_most_likely below for the alignments where that reaches the printed digits)."""
    import numpy as np
    s = np.ascontiguousarray(sets, dtype=np.uint64)
    s = s[s != 0]
    if len(s) == 0:
        return 0.0
    return float(np.add.accumulate(_replay_terms(s))[-1])            # sequential: r[i] = r[i-1] + terms[i]


# ------------------------------------------------------------------ where the reference's own digit hangs on its hash seed
# A codon column with three or more distinct codons is summed in the iteration order of a Python set of strings
# (PolyFastA.py:292-307: list(set(...)), then sum([...]) / len): under the interpreter's default, randomised string hashing,
# an order drawn anew by every run.  Most multisets of synonymous-site fractions give the same float in every order; some do not
# (1/3 + 2/3 + 1. + ... over eight codons: 2.9999999999999996, 3.0 or 3.0000000000000004), and when the alignment's exact
# sum is a two-decimal tie, the reference itself prints x.37 in some runs and x.38 in others (tests/golden G12
# `order_dependent`: the reference under 40 hash seeds).  There is no single answer to be identical to; this implementation
# prints the digits the reference prints MOST OFTEN: the column terms' distributions over summation orders (every order for up to
# seven codons, 4 096 drawn ones above), 512 draws of the whole alignment accumulated in column order like the reference, the
# commonest printed (sites_S, sites_N) pair.  Deterministic: the draws are seeded by the codon sets themselves.
_TERM_POOLS = {}


def _term_pool(vals):
    """Every value sum(order) / k takes over the orders of the float terms `vals` (a sorted tuple), with the frequencies of
    the orders: None when every order gives the same float, else float64[...] to draw from uniformly."""
    import itertools
    import numpy as np
    pool = _TERM_POOLS.get(vals)
    if pool is None and vals not in _TERM_POOLS:
        k = len(vals)
        if k <= 7:
            perms = np.array(list(itertools.permutations(vals)), dtype=np.float64)
        else:
            import zlib
            rng = np.random.default_rng(zlib.crc32(np.array(vals, dtype=np.float64).tobytes()))
            perms = rng.permuted(np.tile(np.array(vals, dtype=np.float64), (4096, 1)), axis=1)
        sums = np.add.accumulate(perms, axis=1)[:, -1] / k          # left to right, like sum() from 0
        pool = None if (sums == sums[0]).all() else sums
        if len(_TERM_POOLS) > 4096:      # (five values, up to 64 of them: a bounded but large family; the common ones come back at once)
            _TERM_POOLS.clear()
        _TERM_POOLS[vals] = pool
    return pool


def syn_sites_most_likely(sets, n_sites, draws=512):
    """count_syn of an alignment on a two-decimal tie, as the reference most often has it: `sets` as for syn_sites_replay (the
    full array, zeros included: they are the columns PolyFastA.py:302 counts as missing), `n_sites` the alignment length.
    Equal to syn_sites_replay(sets) bit for bit when no column's term depends on its summation order."""
    import zlib
    import numpy as np
    global _SYN_FLOAT
    if _SYN_FLOAT is None:
        _SYN_FLOAT = _syn_float_by_codon()
    F = _SYN_FLOAT
    s_all = np.ascontiguousarray(sets, dtype=np.uint64)
    s = s_all[s_all != 0]
    base = syn_sites_replay(s_all)
    if len(s) == 0:
        return base
    k = np.bitwise_count(s)
    loose = []
    for i in np.flatnonzero(k >= 3).tolist():
        m = int(s[i])
        vals = tuple(sorted(float(F[c]) for c in range(64) if (m >> c) & 1))
        pool = _term_pool(vals)
        if pool is not None:
            loose.append((i, pool))
    if not loose:
        return base
    terms = _replay_terms(s)      # the ascending-order terms of every column; the loose columns are redrawn per draw
    rng = np.random.default_rng(zlib.crc32(s_all.tobytes()))
    # only the columns from the first loose one on differ between draws: the prefix is summed once
    first = loose[0][0]
    draws = max(64, min(int(draws), 8_000_000 // max(1, len(terms) - first)))      # bounded memory for alignments of millions of sites
    head = float(np.add.accumulate(terms[:first])[-1]) if first else 0.0
    T = np.tile(terms[first:], (draws, 1))
    for i, pool in loose:
        T[:, i - first] = pool[rng.integers(0, len(pool), size=draws)]
    T[:, 0] = head + T[:, 0] if first else T[:, 0]
    totals = np.add.accumulate(T, axis=1)[:, -1]
    A = int(n_sites) - 3 * int((s_all == 0).sum())
    vals, counts = np.unique(totals, return_counts=True)
    by_pair = {}
    for v, c in zip(vals.tolist(), counts.tolist()):
        pair = (round(v, 2), round(A - v, 2), A - v == 0)      # (a non-synonymous site count of exactly zero is an outcome of its
        e = by_pair.setdefault(pair, [0, 0, v])                 #  own: the reference then dies where it otherwise divides by noise)
        e[0] += c
        if c > e[1]:
            e[1], e[2] = c, v
    base_pair = (round(base, 2), round(A - base, 2), A - base == 0)
    best = max(by_pair.items(), key=lambda kv: (kv[1][0], kv[0] == base_pair, -kv[0][0]))
    return base if best[0] == base_pair else float(best[1][2])
