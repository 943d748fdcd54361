"""CPU oracle for the PolyFastA polymorphism-scan hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``polyfasta_amd/`` may import this
module; it exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` have an independent checker for the HIP
path.  It is a from-scratch restatement of the reference's *algorithm*
(pure-Python loops over dict-of-strings inputs, i.e. small cases only); every
function cites the reference line range it restates (``ref:`` = lines of
``/root/reference/PolyFastA.py``).

Pinning: ``tests/golden/gen_golden.py`` imported the reference in the build
container, checked this module against it on every fixture family, and froze
the reference's outputs under ``tests/golden/*.json``;
``tests/test_oracle_golden.py`` re-checks this module against those files on
every run (no reference needed).

Formulation differences from the product path (deliberate, so the checker is
independent): the product finishes theta_pi from the exact integer
``H = sum_sites (N^2 - sum_j c_j^2)``; the oracle follows the reference's
floating-point ``N/(N-1) * sum_sites (1 - sum_j (c_j/N)^2)``.
"""
from __future__ import annotations

import math
from collections import Counter

# --------------------------------------------------------------------------
# Standard genetic code, derived (not transcribed): TCAG-ordered amino-acid
# string of NCBI translation table 1.
# --------------------------------------------------------------------------
_TCAG = "TCAG"
_AA_TABLE1 = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
AMINO = {
    a + b + c: _AA_TABLE1[16 * i + 4 * j + k]
    for i, a in enumerate(_TCAG)
    for j, b in enumerate(_TCAG)
    for k, c in enumerate(_TCAG)
}
STOP_CODONS = frozenset(c for c, aa in AMINO.items() if aa == "*")  # ref:285
SENSE_CODONS = tuple(sorted(c for c in AMINO if c not in STOP_CODONS))

_IUPAC_OF_SET = {
    frozenset("AG"): "R",
    frozenset("CT"): "Y",
    frozenset("ACGT"): "N",
    frozenset("ACT"): "H",
    frozenset("G"): "G",
}


def codon_annotation(codon: str) -> str:
    """Three-character label of a sense codon: amino acid, size of its
    third-position synonymous family ('0' when the codon is alone) and the
    IUPAC class of that family.  Reproduces the values of the reference's
    lookup table (ref:324-329) from the genetic code itself."""
    aa = AMINO[codon]
    family = frozenset(z for z in "ACGT" if AMINO[codon[:2] + z] == aa)
    size = len(family) if len(family) > 1 else 0
    return f"{aa}{size}{_IUPAC_OF_SET[family]}"


ANNOT = {c: codon_annotation(c) for c in SENSE_CODONS}


def syn_thirds(codon: str) -> int:
    """Number of the nine single-nucleotide neighbours of `codon` that encode
    the same amino acid (stop neighbours never count; stops themselves are 0).
    ``syncodfreq`` (ref:536-557) is this integer divided by three."""
    if codon in STOP_CODONS:
        return 0
    aa = AMINO[codon]
    n = 0
    for p in range(3):
        for b in "ACGT":
            if b != codon[p]:
                nb = codon[:p] + b + codon[p + 1:]
                if AMINO[nb] == aa:
                    n += 1
    return n


def syncodfreq(codon: str) -> float:
    """ref:536-557 — synonymous sites of one codon (KeyError on non-ACGT)."""
    return syn_thirds(codon) / 3


# --------------------------------------------------------------------------
# FASTA reader (ref:227-250)
# --------------------------------------------------------------------------
def parse_fasta_lines(lines):
    """Line loop of ref:231-236 / ref:240-245 on an iterable of text lines
    (already universal-newline translated, each ending in '\\n' except maybe
    the last).  Returns ``None`` where the reference returns 1 (ref:246-248)."""
    records = {}
    current = ""
    for line in lines:
        if line[0] == ">":
            current = line[1:].rstrip()
            records[current] = ""
        elif current != "":
            records[current] += line.rstrip().upper()
    if current == "":
        return None
    return records


def readfasta(path):
    """ref:237-250, file branch."""
    with open(path, "r") as fh:
        return parse_fasta_lines(fh)


# --------------------------------------------------------------------------
# L1 statistics kernel
# --------------------------------------------------------------------------
def getvarsites(d, seqlen):
    """ref:252-261 — positions and columns with more than one distinct
    character; a column is the list of N one-character strings in dict order."""
    names = list(d)
    pos, var = [], []
    for p in range(seqlen):
        column = [d[name][p:p + 1] for name in names]
        if len(set(column)) > 1:
            pos.append(p)
            var.append(column)
    return pos, var


def getsfs(var):
    """ref:274-282 — folded SFS over A/C/G/T alleles only; the bin is the
    second-largest allele count minus one.  IndexError (as in the reference)
    when a column carries fewer than two A/C/G/T alleles."""
    n = len(var[0])
    sfs = [0] * int(n / 2)
    for column in var:
        counts = sorted(
            (column.count(a) for a in set(column) if a in ("A", "C", "G", "T")),
            reverse=True,
        )
        sfs[counts[1] - 1] += 1
    return sfs


def nucleotide_diversity(var):
    """ref:485-492 — N/(N-1) * sum over columns of (1 - sum_j (c_j/N)^2),
    every distinct character being an allele."""
    n = len(var[0])
    total = []
    for column in var:
        freq2 = [(c / n) ** 2 for c in Counter(column).values()]
        total.append(1 - sum(freq2))
    return n / (n - 1) * sum(total)


def wattersons_theta(var):
    """ref:494-497 — S / a1."""
    n = len(var[0])
    a1 = sum(1 / i for i in range(1, n))
    return len(var) / a1


def Dvar(var):
    """ref:522-534 — standard deviation term of Tajima's D."""
    n = len(var[0])
    s = len(var)
    a1 = sum(1.0 / i for i in range(1, n))
    a2 = sum(1.0 / (i ** 2) for i in range(1, n))
    b1 = (n + 1.0) / (3.0 * (n - 1.0))
    b2 = (2.0 * ((n ** 2.0) + n + 3.0)) / (9.0 * n * (n - 1.0))
    c1 = b1 - (1 / a1)
    c2 = b2 - ((n + 2) / (a1 * n)) + (a2 / (a1 ** 2))
    e1 = c1 / a1
    e2 = c2 / ((a1 ** 2) + a2)
    return math.sqrt((e1 * s) + (e2 * s * (s - 1)))


def jukes_cantor_correction(x):
    """ref:499-500."""
    return -0.75 * math.log(1 - (4.0 / 3.0) * x)


def polymorphism(var, seqlen, jc):
    """ref:502-520 — (S, pi/site, theta_w/site, D); D is 'NA' when the
    variance term is exactly zero; a failing JC correction falls back to the
    uncorrected value."""
    if len(var) == 0:
        return 0, 0, 0, "NA"
    pi_total = nucleotide_diversity(var)
    theta_total = wattersons_theta(var)
    try:
        tajima = (pi_total - theta_total) / Dvar(var)
    except ZeroDivisionError:
        tajima = "NA"
    pi_site = pi_total / seqlen
    if jc:
        try:
            pi_site = jukes_cantor_correction(pi_total / seqlen)
        except Exception:
            pi_site = pi_total / seqlen
    return len(var), pi_site, theta_total / seqlen, tajima


# --------------------------------------------------------------------------
# Coding sequences
# --------------------------------------------------------------------------
_THIRD_POS_PAIRS = (  # ref:386-393: annotation sets whose 3rd position is called synonymous
    {"R2R", "H2Y"}, {"L2R", "H2Y"}, {"D2Y", "R2R"}, {"F2Y", "Q2R"},
    {"C2Y", "Q2R"}, {"E2R", "S2Y"}, {"S2Y", "Q2R"},
)
_ILE_EXCLUDED = frozenset(("AAG", "AGG", "GAG", "TTG", "ATT", "ATC"))  # ref:373
_FIRST_POS_BAND = frozenset(("P4N", "L4N", "R4N", "H2Y", "Q2R"))      # ref:401
_ARG_PARTNERS = frozenset(("ATG", "ATA", "ACA", "ACG", "AAA", "AAG"))  # ref:408


def _third_position_is_syn(codons, ann):
    """ref:365-396 — verdict for a varying third position of a two-codon
    column that also varies elsewhere."""
    fam = [a[1:] for a in ann]
    labels = set(ann)
    if "4N" in fam or all(f == "2Y" for f in fam) or all(f == "2R" for f in fam):
        return True
    if "I3H" in labels and ("2Y" in fam or ("2R" in fam and not set(codons) <= _ILE_EXCLUDED)):
        return True
    if "L2R" in labels and "W0G" in labels:
        return True
    if "M0G" in labels and labels & {"R2R", "K2R", "L2R"}:
        return True
    return any(labels <= allowed for allowed in _THIRD_POS_PAIRS)


def _first_position_is_syn(codons, ann):
    """ref:398-411 — verdict for a varying first position of a two-codon
    column that also varies elsewhere."""
    labels = set(ann)
    if labels & {"L2R", "R2R"} and labels & _FIRST_POS_BAND:
        return True
    if "L4N" in labels and ("T0G" in labels or set(codons) & {"TCA", "TCG"}):
        return True
    if "R4N" in labels and set(codons) & _ARG_PARTNERS:
        return True
    return False


def get_syn_nonsyn_cod_sites(cod):
    """ref:319-434 — `cod` is ``[list_of_unique_codons, column_start]``.
    Returns absolute positions called synonymous and non-synonymous."""
    start = cod[1]
    codons = [c for c in cod[0] if c in ANNOT]          # ref:331
    syn, non = [], []
    if len(codons) <= 1:                                 # ref:333
        return syn, non
    varpos = [i for i in range(3) if len({c[i] for c in codons}) > 1]
    ann = [ANNOT[c] for c in codons]
    if len(codons) == 2:
        if len(varpos) == 1:                             # ref:349-358
            p = varpos[0]
            same_label = ann[0] == ann[1]
            leu_or_arg = p == 0 and (set(ann) <= {"L4N", "L2R"} or set(ann) <= {"R4N", "R2R"})
            (syn if same_label or leu_or_arg else non).append(p + start)
        elif len(varpos) > 1:                            # ref:360-414
            if 2 in varpos:
                (syn if _third_position_is_syn(codons, ann) else non).append(2 + start)
            if 0 in varpos:
                (syn if _first_position_is_syn(codons, ann) else non).append(0 + start)
            if 1 in varpos:
                non.append(1 + start)
    else:                                                # ref:415-432
        multiplicity = Counter(ann)
        if len(multiplicity) < len(ann):
            last = varpos[-1]
            distinct_bases = len({c[last] for c in codons})
            if max(multiplicity.values()) >= distinct_bases:
                syn.append(last + start)
            non.extend(p + start for p in varpos[:-1])
        else:
            non.extend(p + start for p in varpos)
    return syn, non


def _is_clean(codon: str) -> bool:
    """ref:301 — ``^[AGTC][AGTC][AGTC]$``."""
    return len(codon) == 3 and all(ch in "ACGT" for ch in codon)


def getvarCDSsites(d, seqlen):
    """ref:284-315 — (synonymous-site count, S positions, N positions,
    number of codon columns holding a stop, 3 x columns without any clean
    codon).  The stop filter of ref:295 never removes anything (it iterates
    over (list, int) tuples), so stops stay in the clean set."""
    names = list(d)
    starts = range(0, seqlen, 3)
    columns = [list(set(d[name][cp:cp + 3] for name in names)) for cp in starts]
    nstops = sum(1 for col in columns if any(c in STOP_CODONS for c in col))
    count_syn = 0.0
    missing = 0
    kept = []
    for col, cp in zip(columns, starts):
        clean = [c for c in col if _is_clean(c)]
        if clean:
            kept.append([clean, cp])
        else:
            missing += 3
    for clean, _ in kept:
        count_syn += sum(syncodfreq(c) for c in clean) / len(clean)
    syn, non = [], []
    for item in kept:
        if len(item[0]) > 1:
            s, n = get_syn_nonsyn_cod_sites(item)
            syn += s
            non += n
    return count_syn, syn, non, nstops, missing


def count_syn_outcomes(d, seqlen, trials=200, seed=0, zero_flag=False):
    """What the reference PRINTS for sites_S / sites_N over its own run-to-run freedom: ref:292 makes each codon column a
    list(set(...)) of strings and ref:307 sums syncodfreq over it with sum([...]) — in the set's iteration order, which
    Python's randomised string hashing draws anew in every run.  Restated with the order drawn explicitly (a shuffle per
    column and trial): Counter {(round(ssites, 2), round(nsites, 2)): trials}.  One entry wherever the reference is
    identical to itself; two on some two-decimal ties (tests/golden G12 `order_dependent`)."""
    import random
    from collections import Counter
    rnd = random.Random(seed)
    names = list(d)
    cols, missing = [], 0
    for cp in range(0, seqlen, 3):
        clean = sorted(c for c in set(d[name][cp:cp + 3] for name in names) if _is_clean(c))
        if clean:
            cols.append([syncodfreq(c) for c in clean])
        else:
            missing += 3
    out = Counter()
    for _ in range(trials):
        count_syn = 0.0
        for vals in cols:
            if len(vals) > 2:
                rnd.shuffle(vals)
            count_syn += sum(vals) / len(vals)
        nsites = (seqlen - missing) - count_syn
        # zero_flag: a non-synonymous site count of exactly 0.0 as an outcome of its own (the reference then dies with
        # ZeroDivisionError where it otherwise divides by the noise of its accumulation)
        out[(round(count_syn, 2), round(nsites, 2)) + ((nsites == 0,) if zero_flag else ())] += 1
    return out


# --------------------------------------------------------------------------
# Row assembly (ref:147-198, the `no_header` closure) as pure functions
# --------------------------------------------------------------------------
NO_VAR = "0,0,0,NA"  # ref:436-437


def result_row(d, seqlen, cds, name, pop, jc):
    """One CSV row exactly as ref:148-198 formats it (without newline)."""
    pos, var = getvarsites(d, seqlen)
    if cds:
        ssites, s, n, nstops, missing = getvarCDSsites(d, seqlen)
        nsites = (seqlen - missing) - ssites
        if len(var) == 0:                                 # ref:151-162
            return f"{name},{round(ssites, 2)},{round(nsites, 2)},{pop},{len(d)},{NO_VAR},{NO_VAR},{0}"
        var_s = [var[pos.index(i)] for i in s if i in pos]
        var_n = [var[pos.index(i)] for i in n if i in pos]
        ps = polymorphism(var_s, ssites, jc)
        pn = polymorphism(var_n, nsites, jc)
        return (f"{name},{round(ssites, 2)},{round(nsites, 2)},{pop},{len(var[0])},"
                f"{ps[0]},{pn[0]},{ps[1]},{pn[1]},{ps[2]},{pn[2]},{ps[3]},{pn[3]},{nstops}")
    if len(var) == 0:                                     # ref:182-189
        return f"{name},{seqlen},{pop},{len(d)},{NO_VAR}"
    p = polymorphism(var, seqlen, jc)
    return f"{name},{seqlen},{pop},{len(var[0])},{p[0]},{p[1]},{p[2]},{p[3]}"


def polymorphism_from_counts(s, h, n, seqlen, jc):
    """ref:502-520 from the integers a scan yields instead of the column lists: `s` variable columns among `n`
    sequences and h = sum over them of (n^2 - sum_j c_j^2), so that ref:485-492's n/(n-1) * sum(1 - sum (c/n)^2)
    is h / (n (n-1)).  The checker's own finishing math for directory-scale tests, where the column-by-column
    functions above would take hours; pinned against them (tests/test_oracle_golden.py)."""
    if s == 0:
        return 0, 0, 0, "NA"
    pi_total = h / (n * (n - 1))
    a1 = sum(1 / i for i in range(1, n))
    theta_total = s / a1
    column = [0] * n
    try:
        tajima = (pi_total - theta_total) / Dvar([column] * s)
    except ZeroDivisionError:
        tajima = "NA"
    pi_site = pi_total / seqlen
    if jc:
        try:
            pi_site = jukes_cantor_correction(pi_total / seqlen)
        except Exception:
            pi_site = pi_total / seqlen
    return s, pi_site, theta_total / seqlen, tajima


def row_from_counts(name, pop, n, seqlen, o, cds, jc):
    """One CSV row (ref:148-198) from the integer summary `o` of a unit (oracle/c_oracle.py: scan): keys S, H and, for
    coding units, seg_S, seg_N, H_S, H_N, nstops, missing_cols, count_syn."""
    if cds:
        ssites = o["count_syn"]
        nsites = (seqlen - 3 * o["missing_cols"]) - ssites
        if o["S"] == 0:
            return f"{name},{round(ssites, 2)},{round(nsites, 2)},{pop},{n},{NO_VAR},{NO_VAR},{0}"
        ps = polymorphism_from_counts(o["seg_S"], o["H_S"], n, ssites, jc)
        pn = polymorphism_from_counts(o["seg_N"], o["H_N"], n, nsites, jc)
        return (f"{name},{round(ssites, 2)},{round(nsites, 2)},{pop},{n},"
                f"{ps[0]},{pn[0]},{ps[1]},{pn[1]},{ps[2]},{pn[2]},{ps[3]},{pn[3]},{o['nstops']}")
    if o["S"] == 0:
        return f"{name},{seqlen},{pop},{n},{NO_VAR}"
    p = polymorphism_from_counts(o["S"], o["H"], n, seqlen, jc)
    return f"{name},{seqlen},{pop},{n},{p[0]},{p[1]},{p[2]},{p[3]}"


def split_pop(d, key):
    """Intended behaviour of ref:125-133 (the reference crashes on Python 3 at
    ref:126): rows whose header contains `key` as a substring, file order."""
    return {h: seq for h, seq in d.items() if key in h}


# --------------------------------------------------------------------------
# Integer summary used to compare with the device records
# --------------------------------------------------------------------------
def integer_summary(d, seqlen):
    """S, H = sum over variable columns of (N^2 - sum_j c_j^2), folded SFS (or
    None where ``getsfs`` raises) and the positions — the integers the HIP
    kernels emit."""
    pos, var = getvarsites(d, seqlen)
    n = len(d)
    h = 0
    for column in var:
        h += n * n - sum(c * c for c in Counter(column).values())
    try:
        sfs = getsfs(var) if var else [0] * int(n / 2)
    except IndexError:
        sfs = None
    return {"S": len(var), "H": h, "sfs": sfs, "pos": pos}
