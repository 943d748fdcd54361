"""Two-decimal rounding ties of sites_S / sites_N (PolyFastA.py:307, :167, :178) — the host side, no GPU: the replay of the
reference's column-order accumulation from per-column codon sets, the tie detector, both formatters, against G12 (rows
frozen from the reference on alignments whose exact synonymous-site sum is a tie)."""
import itertools

import numpy as np
import pytest

from polyfasta_amd import _lib, cli, fasta, stats, synth
from tests import _cases as C


@pytest.fixture(scope="module")
def g12():
    return C.load("g12_rounding_ties.json")


def _units(g12):
    for case in g12["cases"]:
        g = case["gen"]
        m = C.g12_matrix(g)
        hdr = synth.headers_for(g["N"], g.get("pops", 0))
        for key, k in case["keys"].items():
            rows = [i for i, h in enumerate(hdr) if key in h] if key != "NA" else list(range(g["N"]))
            yield g, key, k, m[rows]


def test_fixture_holds_enough_ties_of_every_kind(g12):
    ties = [(c["gen"], key) for c in g12["cases"] for key, k in c["keys"].items() if k["tie"]]
    assert len(ties) >= 20
    gens = [g for g, _ in ties]
    assert any(g.get("pops") for g in gens) and any(g["N"] >= 200 for g in gens) and any(g["N"] >= 1000 for g in gens)
    assert any(g["L"] % 3 for g in gens) and any(g.get("inject") for g in gens) and any(not g.get("plant") for g in gens)


def test_replay_reproduces_the_references_count_syn_bit_for_bit(g12):
    """stats.syn_sites_replay on the codon sets of every G12 unit == the reference's count_syn (repr), whenever the
    reference's own value does not hang on its hash seed; the printed digits always."""
    n = 0
    for g, key, k, sub in _units(g12):
        sets = C.codon_sets_numpy(sub)
        x = stats.syn_sites_replay(sets)
        want = float(k["count_syn"])
        if k["count_syn_order_free"]:
            assert repr(x) == k["count_syn"], (g, key)
            n += 1
        assert round(x, 2) == round(want, 2), (g, key, x, want)
        assert abs(x - want) <= 1e-9
    assert n >= 10


def test_tie_detector_finds_every_tie_and_little_else(g12):
    """from the integer bins alone: every G12 tie is flagged; ordinary alignments are not"""
    bins, L, tie = [], [], []
    for g, key, k, sub in _units(g12):
        bins.append(C.syn_bins_numpy(sub))
        L.append(g["L"])
        tie.append(k["tie"])
    got = stats.syn_tie_candidates(np.array(bins), np.array(L))
    assert got.tolist() == tie
    p = synth.make_params(50, 0.01)
    plain = np.array([C.syn_bins_numpy(synth.synth_alignment(5, a, 3000, p, coding=True)) for a in range(40)])
    assert not stats.syn_tie_candidates(plain, np.full(40, 3000)).any()


def test_rows_with_the_replayed_count_are_the_references_text(g12):
    """cli.format_cds_row and the native formatters, given the replayed count: sites_S / sites_N character for character, the
    rest of the row within the float tolerances — and WITHOUT it the tie rows print the other digit at least once (the
    fixture does pin something)."""
    from oracle import c_oracle
    units, wants, flipped = [], {False: [], True: []}, 0
    for g, key, k, sub in _units(g12):
        o = c_oracle.scan(sub, cds=True)
        rec = (o["H"], o["H_S"], o["H_N"], o["S"], o["sfs_bad"], o["seg_S"], o["seg_N"], o["nstops"], o["missing_cols"], 0, 0)
        syn = C.syn_bins_numpy(sub)
        x = stats.syn_sites_replay(C.codon_sets_numpy(sub))
        units.append(("x.fa", key, len(sub), g["L"], rec, syn, x if k["tie"] else None))
        for jc in (False, True):
            want = k["row_jc" if jc else "row"]
            wants[jc].append(want)
            got = cli.format_cds_row("x.fa", key, len(sub), g["L"], rec, syn, jc, x if k["tie"] else None)
            assert got.split(",")[1:3] == want.split(",")[1:3], (g, key, got, want)
            assert C.row_matches(got, want, True, jc) is None, (g, key, got, want)
        if k["tie"] and cli.format_cds_row("x.fa", key, len(sub), g["L"], rec, syn, False).split(",")[1:3] != k["row"].split(",")[1:3]:
            flipped += 1
    assert flipped >= 3
    for jc in (False, True):
        text = cli.format_cds_rows_native(units, jc)
        assert text == "".join(cli.format_cds_row(*u[:6], jc, u[6]) + "\n" for u in units)
        for got, want in zip(text.rstrip("\n").split("\n"), wants[jc]):
            assert got.split(",")[1:3] == want.split(",")[1:3]
    # the block formatter (what a directory run prints through) and a block that travels between ranks
    meta = np.zeros(len(units), dtype=cli.META_DTYPE)
    meta["fidx"], meta["key"] = np.arange(len(units)), _lib.PFA_NO_KEY
    meta["n_rows"], meta["n_sites"] = [u[2] for u in units], [u[3] for u in units]
    blk = cli.RowBlock(meta, np.array([u[4] for u in units], dtype=_lib.RESULT_DTYPE), np.array([u[5] for u in units], dtype=np.uint32),
                       ssx=np.array([np.nan if u[6] is None else u[6] for u in units]))
    blk2 = cli.RowBlock.from_bytes(blk.to_bytes())
    assert np.array_equal(blk2.ssx, blk.ssx, equal_nan=True)
    both = cli.RowBlock.concat([blk.split_below(7)[1], blk.split_below(7)[0]], True).sorted()
    assert np.array_equal(both.ssx, blk.ssx, equal_nan=True) and np.array_equal(both.syn, blk.syn)
    from polyfasta_amd.ingest import NameTable
    table = NameTable.from_paths([f"d/f{i}.fa" for i in range(len(units))])
    text = cli.format_block_native(both, table, True, False)
    for i, (got, want) in enumerate(zip(text.rstrip("\n").split("\n"), wants[False])):
        assert got.split(",")[1:3] == want.split(",")[1:3] and got.startswith(f"f{i}.fa,")


def test_replay_is_pythons_left_to_right_sum():
    """syn_sites_replay against the reference's loop written out (PolyFastA.py:307) on random codon-set columns, including
    columns with up to 64 codons, empty columns and runs of stop codons"""
    rng = np.random.default_rng(12)
    lib = _lib.load()
    rank = (0, 1, 3, 2)
    f = (0, 1 / 3, 2 / 3, 1., 4 / 3)
    for trial in range(30):
        n = int(rng.integers(1, 4000))
        sets = np.zeros(n, dtype=np.uint64)
        for i in range(n):
            r = rng.random()
            k = 0 if r < 0.05 else 1 if r < 0.8 else 2 if r < 0.93 else int(rng.integers(3, 9)) if r < 0.995 else int(rng.integers(9, 65))
            for c in rng.choice(64, size=k, replace=False).tolist():
                sets[i] |= np.uint64(1) << np.uint64(c)
        total = 0.0
        for s in sets.tolist():
            cods = sorted((c for c in range(64) if (s >> c) & 1), key=lambda c: (rank[c >> 4], rank[(c >> 2) & 3], rank[c & 3]))
            if cods:
                total += sum([f[lib.pfa_codon_syn_thirds(c)] for c in cods]) / len(cods)
        assert stats.syn_sites_replay(sets) == total
    assert stats.syn_sites_replay(np.zeros(0, dtype=np.uint64)) == 0.0 and stats.syn_sites_replay(np.zeros(5, dtype=np.uint64)) == 0.0


def test_exact_ties_need_a_column_of_eight_codons():
    """the arithmetic behind the detector: sums of b_k / (3 k) over k <= 7 are never an odd multiple of 1/200"""
    from fractions import Fraction
    for ks in itertools.combinations(range(1, 8), 3):
        for nums in itertools.product(range(0, 40, 7), repeat=3):
            x = sum(Fraction(b, 3 * k) for b, k in zip(nums, ks))
            assert (200 * x).denominator != 1 or (200 * x).numerator % 2 == 0
    assert (200 * (Fraction(2201) + Fraction(9, 24))).numerator % 2 == 1


def test_replay_equals_the_c_oracles_column_order_sum_on_random_alignments():
    """codon sets (numpy) -> stats.syn_sites_replay == the C oracle's count_syn, the same double, on 300 random coding
    alignments: every alphabet of the random parity tests (gaps, N, IUPAC, stop codons by chance), 1 to 40 rows, lengths
    that are not multiples of 3, mutation rates from 0 to 30 % (columns with many distinct codons)."""
    from oracle import c_oracle
    rng = np.random.default_rng(2025)
    alphabets = [b"ACGT", b"ACGT" * 4 + b"N-", b"ACGTRYKM-N", b"AC", b"ACGT" * 8 + b"acgtn?"]
    n_many = 0
    for trial in range(300):
        n = int(rng.choice([1, 2, 3, 5, 8, 13, 20, 40]))
        L = int(rng.choice([1, 2, 3, 4, 15, 16, 17, 99, 300, 301, 902]))
        alpha = np.frombuffer(alphabets[trial % len(alphabets)], dtype=np.uint8)
        base = rng.choice(alpha, size=L)
        m = np.repeat(base[None, :], n, axis=0)
        hit = rng.random((n, L)) < float(rng.choice([0.0, 0.01, 0.05, 0.3]))
        m[hit] = rng.choice(alpha, size=int(hit.sum()))
        sets = C.codon_sets_numpy(m)
        n_many += int((np.bitwise_count(sets) >= 3).sum())
        want = c_oracle.scan(m, cds=True)["count_syn"]
        got = stats.syn_sites_replay(sets)
        assert got == want, (trial, m.shape, repr(got), repr(want))
    assert n_many > 500


def test_ties_the_reference_does_not_fix_print_what_it_prints_most_often_g15():
    """G15: alignments on a tie whose printed digit follows the reference's own hash seed (its command line under 60 seeds,
    frozen as counts per printed pair).  stats.syn_sites_most_likely on the codon sets gives the pair the reference printed most
    often; the Python oracle's restatement of that freedom (a shuffle per column in place of the set's order) has the same
    outcomes, the same commonest one, and shares within sampling distance of the reference's."""
    import io
    from oracle import polyfasta_oracle as orc
    g15 = C.load("g15_hash_seed_ties.json")
    assert len(g15["cases"]) >= 5
    for case in g15["cases"]:
        d = orc.parse_fasta_lines(io.StringIO(case["text"], newline=None))
        m = fasta.matrix_from_dict(d)      # (one case holds non-ASCII characters: a byte value of its own each)
        L = m.shape[1]
        sets = C.codon_sets_numpy(m)
        x = stats.syn_sites_most_likely(sets, L)
        A = L - 3 * int((sets == 0).sum())
        assert [str(round(x, 2)), str(round(A - x, 2))] == case["most_often"], case["label"]
        assert stats.syn_sites_most_likely(sets, L) == x                      # deterministic
        assert abs(x - stats.syn_sites_replay(sets)) < 1e-12                  # the same sum but for its last bits
        seen = {(a, b): c for a, b, c in case["printed"]}
        assert len(seen) >= 2 and max(seen, key=seen.get) == tuple(case["most_often"])
        out = orc.count_syn_outcomes(d, L, trials=800, seed=15)
        mine = {(str(a), str(b)): c for (a, b), c in out.items()}
        assert set(seen) <= set(mine), (case["label"], seen, mine)
        assert max(mine, key=mine.get) == tuple(case["most_often"])
        share_ref, share_orc = max(seen.values()) / g15["hash_seeds"], max(mine.values()) / 800
        assert abs(share_ref - share_orc) < 0.2, (case["label"], share_ref, share_orc)
    # where no column's term depends on its order, the two functions are the same function
    for g, key, k, sub in _units(C.load("g12_rounding_ties.json")):
        if k["count_syn_order_free"]:
            sets = C.codon_sets_numpy(sub)
            assert stats.syn_sites_most_likely(sets, g["L"]) == stats.syn_sites_replay(sets)


def test_zero_nonsynonymous_sites_die_or_print_like_the_reference_mostly_does_g15():
    """Four sites, one clean codon column of three codons worth 1.0 synonymous site of 1: the non-synonymous site count is zero
    in exact arithmetic, the reference divides by it — by 0.0 in most summation orders (ZeroDivisionError), by 1.1e-16 in the
    others (pi_N = 1.5e16).  The unit is a candidate although it is no rounding tie, and the replay gives the commoner outcome."""
    import io
    from oracle import polyfasta_oracle as orc
    z = C.load("g15_hash_seed_ties.json")["zero_nonsynonymous_sites"]
    d = orc.parse_fasta_lines(io.StringIO(z["text"], newline=None))
    m = fasta.matrix_from_dict(d)
    L = m.shape[1]
    sets = C.codon_sets_numpy(m)
    missing = int((sets == 0).sum())
    assert stats.syn_tie_candidates(C.syn_bins_numpy(m)[None, :], [L], [missing])[0]
    assert not stats.syn_tie_candidates(C.syn_bins_numpy(m)[None, :], [L])[0]          # (no two-decimal tie)
    x = stats.syn_sites_most_likely(sets, L)
    most = max(z["outcomes"], key=lambda o: o["runs"])
    assert most["returncode"] == 1 and most["stderr_tail"].startswith("ZeroDivisionError") and most["runs"] >= 36
    assert (L - 3 * missing) - x == 0.0          # the commoner outcome: a division by exactly zero
    # ordinary units are no candidates on this count
    p = synth.make_params(50, 0.01)
    for a in range(20):
        mm = synth.synth_alignment(7, a, 300, p, coding=True)
        assert not stats.syn_tie_candidates(C.syn_bins_numpy(mm)[None, :], [300], [int((C.codon_sets_numpy(mm) == 0).sum())])[0]


def test_jukes_cantor_singularity_the_oracle_has_the_references_noise_g15():
    """G15 `jc_singularity`: at pi per site = 0.75 the reference prints -0.75 log of its own rounding noise (27.03 / 26.73) or, when
    the noise is nil, the uncorrected 0.75 of its bare `except`.  The Python oracle restates the reference's floating-point evaluation
    of pi (PolyFastA.py:485-492) and lands on one of the rows the reference printed — the very row where the reference has only
    one.  (The product, which finishes pi from exact integers, always prints the fallback: DESIGN.md §8, tests/test_ties_gpu.py.)"""
    import io
    from oracle import polyfasta_oracle as orc
    for case in C.load("g15_hash_seed_ties.json")["jc_singularity"]:
        d = orc.parse_fasta_lines(io.StringIO(case["text"], newline=None))
        L = len(next(iter(d.values())))
        cds = "--cds" in case["flags"]
        row = orc.result_row(d, L, cds, "x.fa", "NA", True)
        assert any(C.row_matches(row, r, cds, True) is None for r, _ in case["rows"]), (case["label"], row, case["rows"])
        # and the exact-integer finish takes the fallback branch
        assert stats.finish_polymorphism(3, 18, 3, 4, True)[1] == 0.75          # 18 / (3 x 2) / 4 = 0.75: log(0), the bare except
