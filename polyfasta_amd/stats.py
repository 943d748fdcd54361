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
